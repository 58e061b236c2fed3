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
#ifdef JSLP_DEBUG_WGLDS
#define WL_MARK(i) do { if (wl_dbg && threadIdx.x == 0) { const unsigned long long _n = __builtin_amdgcn_s_memtime(); atomicAdd(wl_dbg + CNT_DBG + (i), _n - wl_prev); wl_prev = _n; } } while (0)
#define WL_BEGIN(cntp) cnt_t* wl_dbg = (cntp); unsigned long long wl_prev = __builtin_amdgcn_s_memtime()
#else
#define WL_MARK(i) do { } while (0)
#define WL_BEGIN(cntp) do { } while (0)
#endif

struct WgLds {
    double* r0;     // [ld]   cost row (row 0), kept in step with A by the row update
    double* prow;   // [ld]   normalised pivot row of the current pivot
    double* pcol;   // [Hc]   pivot column of the current pivot
    double* rhs;    // [Hc]   column 0 (the authoritative copy during the solve; written back to the slot's mirror at the end)
    int32_t* list;  // [Hc]   rows passing the gate / rows to restore
    int32_t* vibr;  // [Hc]   varIndexByRow / [ld] varIndexByCol: what a pivot swaps and the cycle check records; mirrored to the
    int32_t* vibc;  //        slot's global copies with fire-and-forget stores (a global read here was a full trip per pivot)
    // Branch-and-bound nodes only (k_node_lds / k_node_queue; snapT == nullptr otherwise): a row the node has not written yet
    // still equals the saved root's, so its pivot-column entry comes from the root's TRANSPOSE -- one contiguous column
    // (7.5 KB on Monster_II) shared by every workgroup of the batch instead of H cache lines of the slot, one per row
    uint8_t* cur;         // [Hc] row written by this node (the slot holds its current version)
    const double* snapT;  // column-major copy of the saved root, column stride ldT
    int ldT, Hs;          // Hs = rows of the saved root
    // Copy-on-write nodes (k_node_queue): nothing is restored between nodes -- a row the node has not written yet is READ from
    // the saved root (row-major copy, shared by the whole batch, cache-resident) and written to the slot on its first update;
    // the slot's copy of a row an earlier node wrote and this one did not is stale and never read (it stays flagged dirty; LDS `cur` says what this node wrote)
    const double* snapA;
    bool cow;
    bool preloaded;  // rhs / r0 / vibr / vibc are already in LDS (the queue kernel fills them from the saved root)
    int H_hint, err_hint;  // preloaded only: the height and st->err the simplex starts with (the caller just wrote them)
};
__host__ __device__ __forceinline__ size_t wglds_bytes(int ld, int cap_rows) {
    const size_t hc = ((size_t)cap_rows + 1) & ~(size_t)1;
    return 8 * (2 * (size_t)ld + 2 * hc) + 4 * hc + 4 * hc + 4 * (size_t)ld + hc;
}
__device__ __forceinline__ WgLds wglds_carve(double* base, int ld, int cap_rows) {
    const int hc = (cap_rows + 1) & ~1;
    WgLds L;
    L.r0 = base;
    L.prow = base + ld;
    L.pcol = base + 2 * ld;
    L.rhs = base + 2 * ld + hc;
    L.list = reinterpret_cast<int32_t*>(base + 2 * ld + 2 * hc);
    L.vibr = L.list + hc;
    L.vibc = L.vibr + hc;
    L.cur = reinterpret_cast<uint8_t*>(L.vibc + ld);
    L.snapT = nullptr; L.ldT = 0; L.Hs = 0; L.snapA = nullptr; L.cow = false; L.preloaded = false; L.H_hint = 0; L.err_hint = 0;
    return L;
}

// ---- (key, index) minimum over a workgroup in ONE barrier ---------------------------------------------------------------
// The generic kernels reduce (value, index, batch) triples with 12 dependent ds_bpermute stages and three barriers
// (block_reduce, ~4-5 k cycles per reduction at 16 waves: measured with s_memtime, profiles/r02_wglds_sections.md).  Here a
// candidate is a 64-bit key whose unsigned order IS the selection order plus the index that breaks ties (first index), the
// wave stage is four DPP exchanges inside the 16-lane rows + readlanes across the four rows, and the waves meet in a
// double-buffered LDS array every thread scans itself.
struct KI {
    unsigned long long k;
    int32_t i;
    int32_t pad;
};
#define KI_NONE_KEY (~0ull)
__device__ __forceinline__ KI ki_none() { KI x; x.k = KI_NONE_KEY; x.i = 0x7fffffff; x.pad = 0; return x; }
__device__ __forceinline__ KI ki_min(KI a, KI b) {
    const bool t = b.k < a.k || (b.k == a.k && b.i < a.i);
    KI r; r.k = t ? b.k : a.k; r.i = t ? b.i : a.i; r.pad = 0;
    return r;
}
// doubles -> keys: unsigned order == numeric order (-0 is folded into +0 first: the reference compares them equal and lets
// the first index win); NaN never becomes a candidate (every candidate passed a strict comparison)
__device__ __forceinline__ unsigned long long key_asc(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v + 0.0);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ unsigned long long key_desc(double v) { return ~key_asc(v); }

template <int CTRL>
__device__ __forceinline__ KI ki_dpp(KI x) {
    const int lo = (int)(unsigned)x.k, hi = (int)(unsigned)(x.k >> 32);
    KI y;
    y.k = ((unsigned long long)(unsigned)__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false) << 32) |
          (unsigned long long)(unsigned)__builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    y.i = __builtin_amdgcn_update_dpp(x.i, x.i, CTRL, 0xf, 0xf, false);
    y.pad = 0;
    return y;
}
__device__ __forceinline__ KI ki_readlane(KI x, int lane) {
    KI y;
    y.k = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(x.k >> 32), lane) << 32) |
          (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)x.k, lane);
    y.i = __builtin_amdgcn_readlane(x.i, lane);
    y.pad = 0;
    return y;
}
__device__ __forceinline__ KI ki_wave_min(KI x) {  // result in every lane
    x = ki_min(x, ki_dpp<0xB1>(x));    // quad_perm [1,0,3,2]: lane ^ 1
    x = ki_min(x, ki_dpp<0x4E>(x));    // quad_perm [2,3,0,1]: lane ^ 2
    x = ki_min(x, ki_dpp<0x141>(x));   // row_half_mirror: the other quad of the 8-lane half
    x = ki_min(x, ki_dpp<0x140>(x));   // row_mirror: the other half of the 16-lane row
    KI r = ki_readlane(x, 0);
    r = ki_min(r, ki_readlane(x, 16));
    r = ki_min(r, ki_readlane(x, 32));
    r = ki_min(r, ki_readlane(x, 48));
    return r;
}

#define WGL_HIST 128
struct SmemL {
    Smem g;                 // what suffix_is_square / the counters use
    unsigned long long red_k[2][16];  // cross-wave stage of the reductions (keys / indexes), double-buffered by call parity
    int32_t red_i[2][16];
    int2 hist[WGL_HIST];    // first entries of the cycle-check history (simplex.ts:415-440): the suffix test runs on them
    int32_t n_list;         // length of the gated-row list
    int32_t cyc_hit;        // verdict of the cycle check while the history fits `hist` (wave 0 decides it alone)
};
// Round 5, register diet of the 512-thread batch shapes (80 VGPRs at three workgroups per CU): what these kernels spilled was not
// data but loop-invariant trivia -- zero-extended `tid * 8` offsets, per-thread row / column addresses, threadIdx.y / .z for the
// runtime's __syncthreads_or -- that LLVM hoists out of the node loop and then parks in scratch (13 stores in the prologue, 24
// reloads; tools/kernel_resources.py: 108-156 bytes per lane).  wgl_tid() hands out the thread index behind an opaque barrier:
// whatever is derived from it is recomputed where it is used (two or three VALU instructions) instead of being kept live across a
// whole node, and wgl_block_or() is a workgroup OR that needs nothing but threadIdx.x.  Measured on the Monster_II batch (compact
// read-back): 3.82-3.90 M -> 4.10-4.14 M relaxations/s in tools/wglds_timing.py, 4.52 M -> 5.05 M in bench.py.
__device__ __forceinline__ int wgl_tid() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
__device__ __forceinline__ int wgl_block_or(int pred) {
    __shared__ int wgl_or_flag;
    __syncthreads();  // every thread is past its read of the previous call's verdict
    if (threadIdx.x == 0) wgl_or_flag = 0;
    __syncthreads();
    if (pred) wgl_or_flag = 1;
    __syncthreads();
    return wgl_or_flag;
}
// suffix_is_square (jslp_core.inc.h; checkForCycles, simplex.ts:415-440) on the global history, with the helpers above
__device__ __forceinline__ bool wgl_suffix_is_square(const int2* h, int n) {
    int found = 0;
    const int2 last = h[n - 1];
    for (int L = 1 + wgl_tid(); 2 * L <= n; L += blockDim.x) {
        const int2 a = h[n - 1 - L];
        if (a.x != last.x || a.y != last.y) continue;
        bool eq = true;
        for (int i = 0; i < L - 1; i++) {
            const int2 x = h[n - 2 * L + i], y = h[n - L + i];
            if (x.x != y.x || x.y != y.y) { eq = false; break; }
        }
        if (eq) found = 1;
    }
    return wgl_block_or(found) != 0;
}
// Only the first WGL_SEL threads (four waves, one per SIMD) take part in a selection: a wave-level reduction costs every
// wave its ~60 instructions whether it holds candidates or not, and at 16 waves per workgroup that issue time -- not memory --
// was the pivot's largest cost (block reductions: 6.0 k cycles each at 1024 threads, profiles/r02_wglds_sections.md).
#define WGL_SEL 256
__device__ __forceinline__ KI block_min_ki(KI x, SmemL& sm, int& par) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (w < WGL_SEL / 64) {
        x = ki_wave_min(x);
        // (the lane test is recomputed here, behind an opaque barrier: hoisted out of the pivot loop it becomes a 64-bit SGPR mask
        //  that the 64-VGPR / 78-SGPR build spills, and ROCm 7.2's hipcc reloads only its HIGH half before `s_and_b64 exec, exec, mask`
        //  while the low half's register has meanwhile served as a temporary for the winning index -- DESIGN.md section 8, the
        //  round-2 "64-VGPR fault": with the batch reduction's index 0 in it no lane published, the next reduction read a stale slot)
        int l0 = lane;
        asm volatile("" : "+v"(l0));
        if (l0 == 0) { sm.red_k[par][w] = x.k; sm.red_i[par][w] = x.i; }
    }
    __syncthreads();
    unsigned long long rk = sm.red_k[par][0];
    int ri = sm.red_i[par][0];
#pragma unroll
    for (int i = 1; i < WGL_SEL / 64; i++) {  // (scalars, not structs through references: those end up in scratch behind flat pointers)
        const unsigned long long k = sm.red_k[par][i];
        const int ii = sm.red_i[par][i];
        const bool t = k < rk || (k == rk && ii < ri);
        rk = t ? k : rk;
        ri = t ? ii : ri;
    }
    par ^= 1;  // the next call writes the other buffer: one barrier per reduction is enough
    KI r; r.k = rk; r.i = ri; r.pad = 0;
    return r;
}

// copy one row of ld doubles (a wave; 16 bytes per lane) with four loads in flight per lane instead of a load -> store chain
__device__ __forceinline__ void wglds_copy_row(double2* dst, const double2* src, int ld2, int lane) {
    for (int k0 = lane; k0 < ld2; k0 += 256) {
        double2 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = (k0 + 64 * u < ld2) ? src[k0 + 64 * u] : make_double2(0.0, 0.0);
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (k0 + 64 * u < ld2) dst[k0 + 64 * u] = v[u];
    }
}

// the gated rows of one pivot, one wave per row: row <- row - k * prow on the live columns (simplex.ts:376-387).  The loads
// of UN column pairs are in flight together (a row of Monster_II: two dependent trips instead of eight); column 0 and row 0
// are mirrored in LDS.
// `first`: the node's first write of this row (copy-on-write): every column is read from root_row (the saved root's row) and
// written to the slot, not only the live ones
template <int UN>
// (col_begin: this call covers columns [col_begin, col_begin + 128 * UN): a row is split into such passes so that a pivot with
//  fewer gated rows than waves still keeps every wave busy; col_begin < 0: the whole row)
__device__ __forceinline__ void wglds_update_row(const Ctx& c, const WgLds& L, int r, double k, int pc, double quot, int lane, bool first, const double* root_row,
                                                 int col_begin = -1) {
    const int ld = c.ld;
    double* row = c.A + (long long)r * ld;
    const double* src = first ? root_row : row;
    const int b0 = (col_begin < 0 ? 0 : col_begin) + lane * 2, b1 = col_begin < 0 ? ld : min(ld, col_begin + 128 * UN);
    for (int base = b0; base < b1; base += 128 * UN) {
        double2 a[UN];
        unsigned live = 0;
#pragma unroll
        for (int j = 0; j < UN; j++) {
            const int c0 = base + 128 * j;
            a[j] = make_double2(0.0, 0.0);
            if (c0 < ld) {
                const double2 p = *reinterpret_cast<const double2*>(L.prow + c0);
                if (first || nonzero16(p.x) || nonzero16(p.y) || pc == c0 || pc == c0 + 1) {
                    live |= 1u << j;
                    a[j] = *reinterpret_cast<const double2*>(src + c0);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < UN; j++) {
            if (!(live & (1u << j))) continue;
            const int c0 = base + 128 * j;
            const double2 p = *reinterpret_cast<const double2*>(L.prow + c0);
            double2 x = a[j];
            if (nonzero16(p.x)) x.x = eliminate(x.x, k, p.x);
            if (nonzero16(p.y)) x.y = eliminate(x.y, k, p.y);
            if (pc == c0 || pc == c0 + 1) {
                const double nv = -k / quot;
                if (pc == c0) x.x = nv; else x.y = nv;
            }
            *reinterpret_cast<double2*>(row + c0) = x;
            if (c0 == 0) L.rhs[r] = x.x;
            if (r == 0) *reinterpret_cast<double2*>(L.r0 + c0) = x;
        }
    }
}

// The same for a work item whose column pairs were loaded BEFORE the pivot row was known (round 5, the 1024-thread latency shapes:
// the loads of a pivot's first gated rows leave together with the pivot row's own loads, so a pivot is two dependent global trips --
// pivot column, then pivot row + gated rows -- instead of three).  `a` holds every pair of the pass (the live test needs the
// normalised pivot row, which did not exist when the loads left); what the reference's gate excludes is simply not written back.
template <int UN>
__device__ __forceinline__ void wglds_update_row_preloaded(const Ctx& c, const WgLds& L, int r, double k, int pc, double quot, int lane, bool first,
                                                           int col_begin, const double2 (&a)[UN]) {
    const int ld = c.ld;
    double* row = c.A + (long long)r * ld;
    const int base = col_begin + lane * 2;
#pragma unroll
    for (int j = 0; j < UN; j++) {
        const int c0 = base + 128 * j;
        if (c0 >= ld) continue;
        const double2 p = *reinterpret_cast<const double2*>(L.prow + c0);
        if (!(first || nonzero16(p.x) || nonzero16(p.y) || pc == c0 || pc == c0 + 1)) continue;
        double2 x = a[j];
        if (nonzero16(p.x)) x.x = eliminate(x.x, k, p.x);
        if (nonzero16(p.y)) x.y = eliminate(x.y, k, p.y);
        if (pc == c0 || pc == c0 + 1) {
            const double nv = -k / quot;
            if (pc == c0) x.x = nv; else x.y = nv;
        }
        *reinterpret_cast<double2*>(row + c0) = x;
        if (c0 == 0) L.rhs[r] = x.x;
        if (r == 0) *reinterpret_cast<double2*>(L.r0 + c0) = x;
    }
}

#ifndef WGL_PREFETCH
#define WGL_PREFETCH 1  // 1024-thread builds: every wave's FIRST row-update work item is loaded next to the pivot row (see wglds_update_row_preloaded)
#endif
#ifndef WGL_UN1024
#define WGL_UN1024 (WGL_PREFETCH ? 4 : 8)  // ... of the 1024-thread kernels: 8 = a whole Monster_II row per item (one trip per row); with the prefetch, 4: sixteen waves hold the first 8 rows x 2 passes in 16 registers each
#endif
#ifndef WGL_UN512
#define WGL_UN512 4  // column pairs per lane in flight in one row-update work item of the 512-thread kernels (1024 columns per item)
#endif
#ifndef WGL_PF512
#define WGL_PF512 0  // 1 = the same prefetch in the 512-thread batch shapes: its 16 registers per lane do not fit the 80-VGPR budget of three
                     // workgroups per CU (tools/kernel_resources.py: 20 spilled VGPRs, 36 bytes of scratch in k_node_queue<512, true>)
                     // and measured slower than without it: 3.85-3.94 M against 4.10-4.14 M relaxations/s on the Monster_II batch
                     // (profiles/r05_batch_kernel_register_diet_ab.md)
#endif
#define WGL_KP 2  // pivot-row values a thread keeps in registers across the cycle check (ld <= WGL_KP * threads)

// OPT: the model has optional objectives (their rows stay in the slot's global copy); a build of its own so that the kernels of
// every other model keep their register budget (the extra live values cost the 512-thread queue kernel 10 more VGPR spills)
// PF (round 5, the 1024-thread latency shapes): every wave's first row-update work item is loaded next to the pivot row
#define WGL_OPAQUE_TID(t) do { if (!PF || WGL_PF512) asm volatile("" : "+v"(t)); } while (0)
template <int UN, bool OPT = false, bool PF = false>
__device__ int simplex_wg_lds(const Ctx& c, SmemL& sm, const WgLds& L, int iters_cap) {  // returns what st->err holds at the end (uniform)
    WL_BEGIN(c.cnt);
    DevState* st = c.st;
    int tid = threadIdx.x;
    const int nt = blockDim.x, lane = tid & 63, w = tid >> 6, nw = nt >> 6;
    if (tid == 0) {
        if (L.preloaded) st->err = L.err_hint;  // (begin_simplex keeps a cut error: it re-reads what is stored here, no load of the old value)
        int zero = 0;
        if (!PF || WGL_PF512) asm volatile("" : "+v"(zero));
        begin_simplex(st, iters_cap, zero);
        sm.n_list = 0;
    }
    __syncthreads();
    WGL_OPAQUE_TID(tid);
    const int H = L.preloaded ? L.H_hint : st->H, W = c.W, ld = c.ld;  // the height is fixed during a simplex() call
    const double precision = c.precision;
    double* A = c.A;
    if (!L.preloaded) {  // column 0 and row 0 into LDS (the slot's contiguous RHS mirror when it is valid, else one strided gather)
        const bool mirrored = c.rhs && st->rhs_valid;
        for (int r = tid; r < H; r += nt) L.rhs[r] = mirrored ? c.rhs[r] : A[(long long)r * ld];
        {
            const double* row0 = (L.cow && !L.cur[0]) ? L.snapA : A;
            for (int col = tid; col < ld; col += nt) L.r0[col] = row0[col];
        }
        for (int r = tid; r < H; r += nt) L.vibr[r] = c.vibr[r];
        for (int col = tid; col < W; col += nt) L.vibc[col] = c.vibc[col];
    }
    const int err0 = L.preloaded ? L.err_hint : st->err;
    long long trace_n = (L.preloaded && c.trace_cap == 0) ? 0 : st->trace_n;
    __syncthreads();
    if (err0 != ERR_NONE) {  // a bad cut list: report, do not solve
        if (tid == 0) finish(c);
        return err0;
    }
    WL_MARK(3);
#ifdef JSLP_DEBUG_WGLDS
    if (wl_dbg) {  // micro-costs in this kernel's own geometry (thread 0's clock), 16 repetitions each
        int par0 = 0;
        KI acc = ki_none();
        for (int i = 0; i < 16; i++) { KI x = ki_none(); x.k = key_asc(L.rhs[(tid + i) % H]); x.i = tid; acc = ki_min(acc, block_min_ki(x, sm, par0)); }
        WL_MARK(14);
        for (int i = 0; i < 16; i++) __syncthreads();
        WL_MARK(15);
        double sacc = 0;
        for (int i = 0; i < 16; i++) { for (int r = 1 + tid; r < H; r += nt) sacc += L.rhs[r]; __syncthreads(); }
        WL_MARK(16);
        for (int i = 0; i < 16; i++) { for (int r = tid; r < H; r += nt) L.pcol[r] = A[(long long)r * ld + 1 + i]; __syncthreads(); }
        WL_MARK(17);
        for (int i = 0; i < 16; i++) { for (int col = tid; col < ld; col += nt) L.prow[col] = A[(long long)(1 + i) * ld + col]; __syncthreads(); }
        WL_MARK(18);
        for (int i = 0; i < 16; i++) { for (int col = tid; col < ld; col += nt) c.prow[col] = L.prow[col] + i; __syncthreads(); }
        WL_MARK(19);
        if (acc.k == 1234567 && sacc == 1.5) L.pcol[0] = 1.0;  // keep the loops alive
        __syncthreads();
        for (int r = tid; r < H; r += nt) L.pcol[r] = 0.0;
        __syncthreads();
        WL_MARK(20);
    }
#endif
    const bool row_in_regs = ld <= WGL_KP * nt;
    const double* rootA = L.cow ? L.snapA : A;
    auto rowsrc = [&](int r) -> const double* {  // where row r is read from
        const bool from_root = L.cow && r < L.Hs && !L.cur[r];
        return (from_root ? rootA : A) + (long long)r * ld;
    };
    auto gather_column = [&](int col) {  // the pivot column into LDS
        // Two rows per thread and turn: both addresses first, then BOTH loads, then the LDS stores.  (Written as `L.pcol[r] = ...[r]` in
        // a plain loop the compiler waits for each load before the next turn's: a tableau taller than the workgroup -- Monster_II's 945
        // rows in the 512-thread batch shapes -- paid two dependent trips per gather, 11 k cycles per pivot under load.)
        const double* colT = L.snapT ? L.snapT + (long long)col * L.ldT : nullptr;
        const int Hs1 = L.Hs > 0 ? L.Hs - 1 : 0;
        auto src = [&](int r) -> const double* {  // (branch-free: the flag is read at a clamped index, the pointer is a select)
            const bool untouched = r < L.Hs && L.cur[r < Hs1 ? r : Hs1] == 0;
            const double* mine = A + (long long)r * ld + col;
            if (colT) return untouched ? colT + r : mine;
            return (L.cow && untouched) ? rootA + (long long)r * ld + col : mine;
        };
        for (int r0 = tid; r0 < H; r0 += 2 * nt) {
            const int r1 = r0 + nt, r1c = r1 < H ? r1 : H - 1;  // (a clamped second row: no branch between the two loads)
            const double* p0 = src(r0);
            const double* p1 = src(r1c);
            const double v0 = *p0, v1 = *p1;
            L.pcol[r0] = v0;
            L.pcol[r1c] = v1;  // (unconditional: behind `if (r1 < H)` the compiler sinks the second LOAD into the branch, after the first one's wait; the clamped row's value is that row's own)
        }
    };
    int phase = 1, it1 = 0, it2 = 0, hist_n = 0, iters_left = iters_cap, entered2 = 0, par = 0;
    // outcome: 0 running, 1 optimal, 2 unbounded, 3 cycle, 4 infeasible, 5 iteration cap, 6 history full
    int outcome = 0, unbounded_col = 0;
    while (outcome == 0) {
        if (iters_left <= 0) { outcome = 5; break; }
        WGL_OPAQUE_TID(tid);
        int pr = 0, pc = 0, neg_flag = 0;
        double pv0 = 0.0, pv1 = 0.0;  // my two columns of the pivot row (raw), loaded as early as the row is known
        bool have_pv = false;
        if (phase == 1) {
            // leaving row: most negative RHS below -precision, first index on ties (simplex.ts:39-49)
            double bv = -precision;
            int bi = 0;
            if (tid < WGL_SEL)
                for (int r = 1 + tid; r < H; r += WGL_SEL) {
                    const double v = L.rhs[r];
                    if (v < bv) { bv = v; bi = r; }
                }
            KI x = ki_none();
            if (bi != 0) { x.k = key_asc(bv); x.i = bi; }
            x = block_min_ki(x, sm, par);
            WL_MARK(4);
            if (x.k == KI_NONE_KEY) {  // :51-54 feasible: phase 2 starts in this same iteration with a fresh history (:102)
                phase = 2; entered2 = 1; hist_n = 0;
            } else {
                pr = x.i;
                // entering column: max -cost/coef over unrestricted or coef < -precision (simplex.ts:56-71).  Row pr is the
                // pivot row of this pivot: what is read for the search stays in registers for the normalisation below.
                const double* row = rowsrc(pr);
                double qv = -INFINITY;
                int qi = 0;
                if (row_in_regs) {
                    pv0 = tid < ld ? row[tid] : 0.0;
                    pv1 = tid + nt < ld ? row[tid + nt] : 0.0;
                    have_pv = true;
                    if (tid < ld) L.prow[tid] = pv0;          // (raw: the normalisation below overwrites it)
                    if (tid + nt < ld) L.prow[tid + nt] = pv1;
                } else {
                    for (int col = tid; col < ld; col += nt) L.prow[col] = row[col];
                }
                __syncthreads();
                if (tid < WGL_SEL)
                    for (int col = 1 + tid; col < W; col += WGL_SEL) {
                        const double coef = L.prow[col];
                        const bool un = c.has_unr && c.unr[L.vibc[col]] != 0;
                        if (un || coef < -precision) {
                            const double quo = -L.r0[col] / coef;
                            if (qv < quo) { qv = quo; qi = col; }
                        }
                    }
                KI q = ki_none();
                if (qi != 0) { q.k = key_desc(qv); q.i = qi; }
                q = block_min_ki(q, sm, par);
                if (q.k == KI_NONE_KEY) { outcome = 4; break; }  // :73-76 infeasible
                pc = q.i;
                gather_column(pc);
                __syncthreads();
                WL_MARK(13);
            }
        }
        if (phase == 2) {
            // Dantzig pricing with the reference's batch rule (simplex.ts:118-219) on the LDS cost row: the first
            // batch holding a candidate wins, inside it the largest value, first index on ties
            double ev = precision;
            int ei = 0, eb = 0;
            for (int col = 1 + tid; col < W && tid < WGL_SEL; col += WGL_SEL) {
                const double rc = L.r0[col];
                const bool un = c.has_unr && c.unr[L.vibc[col]] != 0;
                const int b = c.use_partial ? (col - 1) / c.batch : 0;
                const double val = (un && rc < 0) ? -rc : rc;
                if (val > precision) {
                    const bool take = ei == 0 || b < eb || (b == eb && val > ev);  // my columns ascend: ties keep the earlier one
                    ev = take ? val : ev;
                    ei = take ? col : ei;
                    eb = take ? b : eb;
                }
            }
            if (c.use_partial) {  // which batch?
                KI b = ki_none();
                if (ei != 0) { b.k = (unsigned long long)eb; b.i = 0; }
                b = block_min_ki(b, sm, par);
                if (b.k == KI_NONE_KEY) ei = 0;
                else if (ei != 0 && (unsigned long long)eb != b.k) ei = 0;  // my best sits in a later batch
            }
            KI e = ki_none();
            if (ei != 0) { e.k = key_desc(ev); e.i = ei; }
            e = block_min_ki(e, sm, par);
            WL_MARK(5);
            int opt_row = -1;  // >= 0: the entering column is named by that optional objective (simplex.ts:221-263)
            for (int o = 0; OPT && e.k == KI_NONE_KEY && o < c.n_opt; o++) {
                // no column prices out on the main row (nor on the earlier objectives): objective o breaks the tie among the columns
                // whose reduced cost is within +-precision on all of them.  The objective rows live in the slot's global copy
                // (n_opt x ld doubles, cache-resident): a rare path
                double xv = precision;
                int xi = 0;
                for (int col = 1 + tid; col < W && tid < WGL_SEL; col += WGL_SEL) {
                    const double rc0 = L.r0[col];
                    bool deferred = -precision < rc0 && rc0 < precision;
                    for (int q = 0; deferred && q < o; q++) {
                        const double rq = c.oo[(long long)q * ld + col];
                        deferred = -precision < rq && rq < precision;
                    }
                    if (!deferred) continue;
                    const double rc = c.oo[(long long)o * ld + col];
                    if (-precision < rc && rc < precision) continue;
                    const bool un = c.has_unr && c.unr[L.vibc[col]] != 0;
                    const double val = (un && rc < 0) ? -rc : rc;
                    const bool take = val > xv;  // strict: my columns ascend, ties keep the earlier one
                    xv = take ? val : xv;
                    xi = take ? col : xi;
                }
                KI x = ki_none();
                if (xi != 0) { x.k = key_desc(xv); x.i = xi; }
                e = block_min_ki(x, sm, par);
                if (e.k != KI_NONE_KEY) opt_row = o;
            }
            if (e.k == KI_NONE_KEY) { outcome = 1; break; }  // optimal (simplex.ts:265-269)
            pc = e.i;
            {
                const double rc = (!OPT || opt_row < 0) ? L.r0[pc] : c.oo[(long long)opt_row * ld + pc];
                const bool un = c.has_unr && c.unr[L.vibc[pc]] != 0;
                neg_flag = (un && rc < 0) ? 1 : 0;
            }
            // ratio test (simplex.ts:271-296) in its order-free form: the first degenerate row wins outright (key 0), else the
            // first-index argmin of the accepted quotients; the strided column gather is the pivot's first global trip
            double mv = INFINITY;
            int mi = 0, rdeg = 0x7fffffff;
            gather_column(pc);
            __syncthreads();
            for (int r = 1 + tid; r < H && tid < WGL_SEL; r += WGL_SEL) {
                const double colv = L.pcol[r];
                const double rhs = L.rhs[r];
                if (-precision < colv && colv < precision) continue;
                if (colv > 0 && precision > rhs && rhs > -precision) {
                    if (r < rdeg) rdeg = r;
                    continue;
                }
                const double quo = neg_flag ? -rhs / colv : rhs / colv;
                if (quo > precision && mv > quo) { mv = quo; mi = r; }
            }
            KI m = ki_none();
            if (rdeg != 0x7fffffff) { m.k = 0; m.i = rdeg; }
            else if (mi != 0) { m.k = key_asc(mv); m.i = mi; }
            m = block_min_ki(m, sm, par);
            WL_MARK(6);
            if (m.k == KI_NONE_KEY) { outcome = 2; unbounded_col = pc; break; }  // unbounded (simplex.ts:298-303)
            pr = m.i;
        }
        // ---- the pivot (pr, pc) is chosen.  Start its global reads now: my columns of the pivot row, and (thread 0) the two
        //      map entries the pivot swaps; the cycle check and the row-gate compaction run while they are in flight ---------
        const double* prow_src = rowsrc(pr);  // (read before anyone marks the row as written)
        if (row_in_regs && !have_pv) {
            const double* row = prow_src;
            pv0 = tid < ld ? row[tid] : 0.0;
            pv1 = tid + nt < ld ? row[tid + nt] : 0.0;
            have_pv = true;
        }
        int leaving = 0, entering = 0;
        if (tid == 0) { leaving = L.vibr[pr]; entering = L.vibc[pc]; }
        // the rows that pass the reference's gate (simplex.ts:370-375), compacted into the LDS list (a Monster_II pivot: ~10 of
        // 945); "any row at all" is what decides the lazy zeroing of tiny pivot-row entries (:381-383)
        int n_gated = 0;
        for (int r = tid; r < H; r += nt) {
            if (r != pr && nonzero16(L.pcol[r])) {
                const bool first = L.cow && r < L.Hs && !L.cur[r];  // copy-on-write: this pivot brings the row into the slot
                L.list[atomicAdd(&sm.n_list, 1)] = r | (first ? 0x40000000 : 0);  // the list holds every row: it cannot overflow
                c.dirty[r] = 1;
                L.cur[r] = 1;
                n_gated += 1;
            }
        }
        // cycle check (simplex.ts:78-93 / 305-320): append first, test, stop WITHOUT pivoting on a hit
        if (c.check_cycles) {
            if (hist_n >= c.hist_cap) { outcome = 6; break; }
            // Round 5: while the history fits its LDS copy (the first WGL_HIST pairs: every node of a tree, every small LP) the suffix test
            // is WAVE 0's alone -- lane L tests block length L (and L + 64) on the LDS copy, one ballot, the verdict rides the barrier
            // that completes the gated-row list anyway.  The block-wide form (suffix_is_square: every thread through the test's loop
            // bounds, then a __syncthreads_or) cost 2.7 k cycles per pivot of a single node and 9 k per pivot in the batch kernel
            // (profiles/r02_wglds_sections.md: 9 % of a node either way) for histories of five or six pairs.
            const bool short_hist = hist_n < WGL_HIST;
            if (w == 0) {
                if (tid == 0) {
                    const int2 pair = make_int2(leaving, entering);
                    c.hist[hist_n] = pair;  // the host rebuilds the reference's [start, length] message from the global copy
                    if (short_hist) sm.hist[hist_n] = pair;
                }
                if (short_hist) {
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): thread 0's LDS store is this wave's own
                    const int n1 = hist_n + 1;
                    const int2 last = sm.hist[n1 - 1];
                    int found = 0;
                    for (int L = 1 + (wgl_tid() & 63); 2 * L <= n1; L += 64) {
                        const int2 a = sm.hist[n1 - 1 - L];
                        if (a.x != last.x || a.y != last.y) continue;
                        bool eq = true;
                        for (int i = 0; i < L - 1; i++) {
                            const int2 x = sm.hist[n1 - 2 * L + i], y = sm.hist[n1 - L + i];
                            if (x.x != y.x || x.y != y.y) { eq = false; break; }
                        }
                        if (eq) found = 1;
                    }
                    const unsigned long long hit = __ballot(found != 0);
                    if (lane == 0) sm.cyc_hit = hit != 0ull ? 1 : 0;
                }
            }
            __syncthreads();
            hist_n += 1;
            const bool cycle = short_hist ? sm.cyc_hit != 0 : wgl_suffix_is_square(c.hist, hist_n);
            if (cycle) { outcome = 3; break; }
        } else {
            __syncthreads();  // the list is complete
        }
        WL_MARK(7);
        const int n = sm.n_list;
        const bool anyrow = n > 0;
        // ---- round 5: my wave's first row-update work item leaves NOW, next to the pivot row's loads (issued above): by the time the
        //      normalised pivot row is in LDS the gated rows are in registers too.  Every column pair of the pass is loaded (which ones are
        //      live is only known with the pivot row); the rows are not written by anybody before the update pass below reads them.
        constexpr bool PREFETCH = WGL_PREFETCH != 0 && PF;
        const int passes_pf = (ld + 128 * UN - 1) / (128 * UN);
        double2 pf[UN];
        int pf_r = -1, pf_col0 = 0;
        bool pf_first = false;
        if (PREFETCH && w < n * passes_pf) {
            const int i = w / passes_pf, ps = w - i * passes_pf;
            const int entry = L.list[i];
            pf_r = entry & 0x3fffffff;
            pf_first = (entry & 0x40000000) != 0;
            pf_col0 = ps * 128 * UN;
            const double* src = (pf_first ? rootA : A) + (long long)pf_r * ld;
#pragma unroll
            for (int j = 0; j < UN; j++) {
                const int c0 = pf_col0 + lane * 2 + 128 * j;
                pf[j] = c0 < ld ? *reinterpret_cast<const double2*>(src + c0) : make_double2(0.0, 0.0);
            }
        }
        // ---- pivot row (simplex.ts:352-364) from the registers (or from memory when the row is wider than they hold) -----------
        const double quot = L.pcol[pr];  // = A[pr, pc] (:335)
        double* prow_A = A + (long long)pr * ld;
        int n_cols = 0;
        if (have_pv) {
            auto normalise = [&](int col, double val) {
                if (col < ld) {
                    double v = 0.0;
                    if (col < W) {
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
            };
            normalise(tid, pv0);
            normalise(tid + nt, pv1);
        } else {
            for (int col = tid; col < ld; col += nt) {
                double v = 0.0;
                if (col < W) {
                    const double val = prow_src[col];
                    const bool innz = nonzero16(val);
                    v = innz ? val / quot : 0.0;
                    if (col == pc) v = 1.0 / quot;
                    if (innz && anyrow && !nonzero16(v) && v != 0.0) v = 0.0;
                    prow_A[col] = v;
                    if (col == 0) L.rhs[pr] = v;
                    n_cols += (nonzero16(v) || col == pc) ? 1 : 0;
                }
                L.prow[col] = v;
            }
        }
        if (tid == 0) {  // :339-349
            c.vibr[pr] = entering;
            c.vibc[pc] = leaving;
            L.vibr[pr] = entering;
            L.vibc[pc] = leaving;
            c.rbv[entering] = pr;
            c.rbv[leaving] = -1;
            c.cbv[entering] = -1;
            c.cbv[leaving] = pc;
            if (trace_n < c.trace_cap) c.trace[trace_n] = make_int2(pr, pc);
            c.dirty[pr] = 1;
            L.cur[pr] = 1;
        }
        trace_n += 1;
        if (phase == 1) it1 += 1; else it2 += 1;
        iters_left -= 1;
        if (c.cnt) {  // work counters (uniform branch): cells simplex.ts:376-387 touches = gated rows x live pivot-row columns
            for (int off = 32; off > 0; off >>= 1) n_cols += __shfl_down(n_cols, off, 64);
            __syncthreads();
            if (tid == 0) sm.g.flag = 0;
            __syncthreads();
            if (lane == 0) atomicAdd(&sm.g.flag, n_cols);
            __syncthreads();
            if (tid == 0) {
                atomicAdd(c.cnt + CNT_CELLS, (cnt_t)n * (cnt_t)sm.g.flag);
                atomicAdd(c.cnt + CNT_ROWS, (cnt_t)n);
            }
        }
        (void)n_gated;
        __syncthreads();  // L.prow complete
        // optional objectives (simplex.ts:394-412): the same elimination with exact `!== 0` tests, on the FINAL pivot row; the rows
        // live in the slot's global copy (the loads below are in flight while the row updates run)
        for (int o = 0; OPT && o < c.n_opt; o++) {
            double* rc = c.oo + (long long)o * ld;
            const double coefficient = rc[pc];  // every thread reads it ...
            __syncthreads();                    // ... before anyone overwrites rc[pc]
            if (coefficient != 0.0) {
                for (int col = tid; col < W; col += nt) {
                    if (col == pc) { rc[col] = -coefficient / quot; continue; }
                    const double v0 = L.prow[col];
                    if (v0 != 0.0) rc[col] = eliminate(rc[col], coefficient, v0);
                }
            }
        }
        WL_MARK(8);
        {   // work items = (gated row, pass of UN column pairs per lane = 128 * UN columns).  Every item is one dependent memory
            // trip for its wave, so what counts under load is trips per wave: measured on the Monster_II batch (8 waves, ~10 gated
            // rows per pivot, 928 columns) row updates cost 98 k cycles per node with UN = 4 as whole rows (2 trips x 2 rounds),
            // 93 k with 512-column items (this), 133 k with 256-column items (5 rounds); UN = 8 (one trip per row) does not fit the
            // 80-VGPR budget of the batch shape (224 bytes of scratch) and that build lost pivots -- not used
            const int passes = (ld + 128 * UN - 1) / (128 * UN);
            int it = w;
            if (PREFETCH && pf_r >= 0) {  // the item loaded ahead (same item `w`: same row, same pass)
                wglds_update_row_preloaded<UN>(c, L, pf_r, L.pcol[pf_r], pc, quot, lane, pf_first, pf_col0, pf);
                it += nw;
            }
            for (; it < n * passes; it += nw) {
                const int i = it / passes, ps = it - i * passes;
                const int entry = L.list[i], r = entry & 0x3fffffff;
                wglds_update_row<UN>(c, L, r, L.pcol[r], pc, quot, lane, (entry & 0x40000000) != 0, rootA + (long long)r * ld, ps * 128 * UN);
            }
        }
        if (tid == 0) sm.n_list = 0;  // for the next pivot (several barriers away from its first use)
        __syncthreads();
        WL_MARK(10);
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
        if (outcome == 2) { st->bounded = 0; st->unbounded_var = L.vibc[unbounded_col]; }
        if (outcome == 3) { st->cycle_phase = phase; st->feasible = 0; }
        if (outcome == 4) st->feasible = 0;
        if (outcome == 5) st->err = ERR_ITER_LIMIT;
        if (outcome == 6) st->err = ERR_HIST_FULL;
        st->status = ST_DONE;
        st->do_pivot = 0;
        st->obj_cell = L.r0[0];
    }
    __syncthreads();
    WL_MARK(11);
    return outcome == 5 ? (int)ERR_ITER_LIMIT : outcome == 6 ? (int)ERR_HIST_FULL : (int)ERR_NONE;
}

// addCutConstraints (cutting-strategies.ts:16-72) with one WAVE per cut row (add_cuts_slot builds them one after the other:
// five dependent global trips per node of a Monster_II tree); the slack bookkeeping stays sequential (getNewElementIndex)
// (rows_src: where the rows of the root are read from -- the slot, or the saved root itself for copy-on-write nodes)
// (maps_rbv / maps_cbv: where rowByVarIndex / colByVarIndex of the cut variables are read from -- the slot's maps or the saved
//  root's; Lm: LDS mirrors of column 0 and the row map to keep in step, or nullptr)
// Returns (uniformly) the slot's error code after the cuts when the caller passed H_known and lei_known (ERR_NONE, ERR_CAPACITY,
// ERR_CUT_ARG: what st->err holds), else -1 (read st->err).
__device__ __forceinline__ int add_cuts_waves(const Slots& s, const Cuts& cuts, int slot, int node, int cap_rows, const double* rows_src = nullptr,
                                               const int32_t* maps_rbv = nullptr, const int32_t* maps_cbv = nullptr, const WgLds* Lm = nullptr,
                                               int H_known = -1, int lei_known = -1, int a_known = -1, int n_known = -1) {
    DevState* st = s.st + slot;
    double* A = s.A + (long long)slot * s.A_stride;
    double* rhs = s.rhs + (long long)slot * s.pcol_stride;
    int32_t* vibr = s.vibr + (long long)slot * s.vibr_stride;
    int32_t* rbv = s.rbv + (long long)slot * s.idx_stride;
    int32_t* cbv = s.cbv + (long long)slot * s.idx_stride;
    // (a_known / n_known: the node's slice of the cut lists, read by the caller next to its first loads -- one trip less in this chain)
    const int a = a_known >= 0 ? a_known : cuts.offs[node], n = a_known >= 0 ? n_known : cuts.offs[node + 1] - a;
    // (H_known / lei_known: the caller set st->H / st->last_element_index a moment ago and says what to; re-reading them is a
    //  global round trip each, and the slack loop below used to pay one per cut)
    const int H = H_known >= 0 ? H_known : st->H, W = s.W, ld = s.ld;
    const int tid_c = wgl_tid(), lane = tid_c & 63, w = tid_c >> 6, nw = blockDim.x >> 6;
    if (H + n > cap_rows) {
        if (threadIdx.x == 0) st->err = ERR_CAPACITY;
        return (H_known >= 0 && lei_known >= 0) ? (int)ERR_CAPACITY : -1;
    }
    int my_bad = 0;
    for (int h = w; h < n; h += nw) {
        const int vi = cuts.var[a + h];
        const double sign = cuts.type[a + h] == 0 ? -1.0 : 1.0;  // "min" -> -1 (:41)
        const double value = cuts.value[a + h];
        double* cut = A + (long long)(H + h) * ld;
        const int var_row = (vi >= 0 && vi < s.idx_stride) ? (maps_rbv ? maps_rbv : rbv)[vi] : -2;
        const int var_col = (vi >= 0 && vi < s.idx_stride) ? (maps_cbv ? maps_cbv : cbv)[vi] : -1;
        if (var_row == -2 || (var_row == -1 && var_col < 0)) {
            if (lane == 0) st->err = ERR_CUT_ARG;
            my_bad = 1;
            continue;
        }
        if (var_row == -1) {  // non-basic variable: unit row (:46-53)
            for (int col = lane; col < ld; col += 64) {
                double v = 0.0;
                if (col == 0) { v = sign * value; rhs[H + h] = v; if (Lm) Lm->rhs[H + h] = v; }
                else if (col == var_col) v = sign;
                cut[col] = v;
            }
        } else {  // basic variable: negated copy of its row (:54-62)
            // (column pairs, four 16-byte loads in flight per lane: 512 columns per trip -- as four doubles per lane and turn a Monster_II
            //  row was four dependent trips of the seven this function's chain had, now two; eight pairs in flight would be one, but their
            //  32 registers put the batch kernels back into scratch; ld is even, columns >= W read as 0)
            const double2* src2 = reinterpret_cast<const double2*>((rows_src ? rows_src : A) + (long long)var_row * ld);
            double2* cut2 = reinterpret_cast<double2*>(cut);
            const int ld2 = ld >> 1;
            for (int q0 = lane; q0 < ld2; q0 += 256) {
                double2 x[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const int q = q0 + 64 * u; x[u] = src2[q < ld2 ? q : ld2 - 1]; }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int q = q0 + 64 * u, col = 2 * q;
                    if (q >= ld2) continue;
                    double2 v;
                    v.x = col < W ? -sign * x[u].x : 0.0;
                    v.y = col + 1 < W ? -sign * x[u].y : 0.0;
                    if (q == 0) { v.x = sign * (value - x[u].x); rhs[H + h] = v.x; if (Lm) Lm->rhs[H + h] = v.x; }
                    cut2[q] = v;
                }
            }
        }
    }
    const int bad = wgl_block_or(my_bad);  // (a bad cut list: every thread knows, nobody re-reads st->err)
    if (threadIdx.x == 0 && !bad) {
        int lei = lei_known >= 0 ? lei_known : st->last_element_index;
        for (int h = 0; h < n; h++) {  // getNewElementIndex + map updates (:64-69)
            const int slack = lei++;
            if (slack >= s.idx_stride) { st->err = ERR_CAPACITY; break; }
            vibr[H + h] = slack;
            if (Lm) Lm->vibr[H + h] = slack;
            rbv[slack] = H + h;
            cbv[slack] = -1;
        }
        st->last_element_index = lei;
        st->H = H + n;
    }
    if (H_known < 0 || lei_known < 0) return -1;
    if (bad) return (int)ERR_CUT_ARG;
    return lei_known + n > s.idx_stride ? (int)ERR_CAPACITY : (int)ERR_NONE;
}

// simplex() of slots [first_slot, first_slot + gridDim.x): the LDS twin of k_simplex_wg
template <int THREADS, bool OPT = false>
__global__ void __launch_bounds__(THREADS) k_simplex_lds(Slots s, int first_slot, int check_cycles, int iters_cap, int cap_rows) {
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
    __shared__ SmemL sm;
    const Ctx c = slot_ctx(s, first_slot + blockIdx.x, check_cycles);
    const WgLds L = wglds_carve(lds_dyn, s.ld, cap_rows);
    simplex_wg_lds<(THREADS >= 1024 ? WGL_UN1024 : WGL_UN512), OPT, (THREADS >= 1024 || WGL_PF512)>(c, sm, L, iters_cap);
}

// The LDS twin of k_node_wg: ONE branch-and-bound child per workgroup in ONE launch -- restore of the rows the previous node
// dirtied, the index maps, addCutConstraints, simplex() and the read-back (see k_node_wg for the contract).
#ifndef JSLP_NODE512_WAVES
// waves per SIMD the 512-thread batch shape is compiled for.  6 = three workgroups per CU, 80 VGPRs: measured (round 2)
// 1.87 M relaxations/s on the Monster_II batch against 1.79 M for 8 (64 VGPRs, 36 bytes of scratch per lane); with the LDS
// this kernel declares (Monster_II: 47.5 KB dynamic + 1.7 KB static per workgroup) a fourth workgroup does not fit a CU anyway.
// Round 5: none of the three k_node_queue<512> builds touches scratch any more (0 spilled VGPRs at 75-80; they were 26-38 with
// 108-156 bytes per lane -- see wgl_tid); at 8 waves per SIMD the copy-on-write build would still spill 6.
#define JSLP_NODE512_WAVES 6
#endif
// Read-back of a node the LDS kernel just solved: column 0 and the row map are still in LDS (no load of what this workgroup
// stored a moment ago).  Compact form (out_stride < 0): the row of each watched variable is found by scattering the rows
// through watch_pos (variable index -> position in the watched list, -1 elsewhere) into the LDS list.
__device__ __forceinline__ void gather_slot_lds(const Slots& s, const WgLds& L, int slot, double* rhs, int32_t* rows, DevState* states,
                                                int out_stride, int o, int H_known = -1) {
    const DevState* st = s.st + slot;
    const int H = H_known >= 0 ? H_known : st->H, tid = wgl_tid();
    if (out_stride < 0) {
        const int n = -out_stride < s.n_watch ? -out_stride : s.n_watch;
        if (s.watch_pos) {
            for (int i = tid; i < n; i += blockDim.x) L.list[i] = -1;
            __syncthreads();
            for (int r0 = 1 + tid; r0 < H; r0 += 2 * (int)blockDim.x) {  // (both look-ups of a turn in flight together: clamped, unconditional)
                const int r1 = r0 + (int)blockDim.x, r1c = min(r1, H - 1);
                const int v0 = L.vibr[r0], v1 = L.vibr[r1c];
                const bool ok0 = v0 >= 0 && v0 < s.idx_stride, ok1 = v1 >= 0 && v1 < s.idx_stride;
                const int q0 = s.watch_pos[ok0 ? v0 : 0], q1 = s.watch_pos[ok1 ? v1 : 0];
                const int p0 = ok0 ? q0 : -1, p1 = ok1 ? q1 : -1;
                if (p0 >= 0 && p0 < n) L.list[p0] = r0;
                if (p1 >= 0 && p1 < n) L.list[p1] = r1c;
            }
        } else {
            // a variable listed twice has no single position: every listed entry looks its row up in the LDS row map (the global
            // maps of a copy-on-write slot are stale, so gather_slot() is not an option here)
            for (int i = tid; i < n; i += blockDim.x) {
                const int v = s.watch[i];
                int row = -1;
                for (int r = 1; r < H; r++) row = L.vibr[r] == v ? r : row;
                L.list[i] = row;
            }
        }
        __syncthreads();
        for (int i = tid; i < n; i += blockDim.x) {
            const int r = L.list[i];
            if (rows) rows[(long long)o * (-out_stride) + i] = r;
            if (rhs) rhs[(long long)o * (-out_stride) + i] = r > 0 ? L.rhs[r] : 0.0;
        }
    } else if ((out_stride & 3) == 0) {
        // the node's slices start 16-byte aligned: 16-byte stores of the RHS column, 8-byte stores of the row map (4-byte stores
        // into pinned host memory are several times slower per byte)
        if (rhs) {
            double2* dst = reinterpret_cast<double2*>(rhs + (long long)o * out_stride);
            const double2* src = reinterpret_cast<const double2*>(L.rhs);
            for (int k = tid; 2 * k < H; k += blockDim.x) {
                double2 v = src[k];
                if (2 * k + 1 >= H) v.y = 0.0;
                dst[k] = v;
            }
        }
        if (rows) {
            int2* dst = reinterpret_cast<int2*>(rows + (long long)o * out_stride);
            const int2* src = reinterpret_cast<const int2*>(L.vibr);
            for (int k = tid; 2 * k < H; k += blockDim.x) {
                int2 v = src[k];
                if (2 * k + 1 >= H) v.y = -1;
                dst[k] = v;
            }
        }
    } else {
        for (int r = tid; r < H; r += blockDim.x) {
            if (rhs) rhs[(long long)o * out_stride + r] = L.rhs[r];
            if (rows) rows[(long long)o * out_stride + r] = L.vibr[r];
        }
    }
    if (tid == 0) states[o] = *st;
}

// one node (= restore + cuts + simplex + read-back) of slot `slot`; false = the slot is not in sync with the snapshot
template <int THREADS, bool COW = false, bool OPT = false>
__device__ __forceinline__ bool node_lds_run(const Slots& s, const Snapshot& snap, const Cuts& cuts, SmemL& sm, const WgLds& L, int slot, int node, int o,
                                             int check_cycles, int iters_cap, int cap_rows, double* rhs_out, int32_t* rows_out,
                                             DevState* state_out, int out_stride) {
    WL_BEGIN(s.cnt);
    DevState* st = s.st + slot;
    const int tid = THREADS == 512 ? wgl_tid() : (int)threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
    const int gen = s.st[0].s_gen, H = s.st[0].s_H, ld2 = s.ld / 2;  // every slot shares slot 0's snapshot scalars
    const int cut_a = cuts.offs[node], cut_n = cuts.offs[node + 1] - cut_a;  // (in flight with the loads above: add_cuts_waves starts with them)
    const int lei0 = s.st[0].s_last_element_index;
    if (gen == 0 || st->gen != gen) {  // must not happen (host bookkeeping): refuse rather than restore wrongly
        if (tid == 0) { st->err = ERR_NOT_SYNCED; st->status = ST_DONE; state_out[o] = *st; }
        return false;
    }
    double* A = s.A + (long long)slot * s.A_stride;
    uint8_t* dirty = s.dirty + (long long)slot * s.pcol_stride;
    double* rhs = s.rhs + (long long)slot * s.pcol_stride;
    int32_t* vibr = s.vibr + (long long)slot * s.vibr_stride;
    int32_t* vibc = s.vibc + (long long)slot * s.vibc_stride;
    int32_t* rbv = s.rbv + (long long)slot * s.idx_stride;
    int32_t* cbv = s.cbv + (long long)slot * s.idx_stride;
    int cut_err = (int)ERR_NONE;
    auto restore_maps = [&]() {  // the index maps: all loads of a pass issued before its stores
        const int nt4 = blockDim.x * 4;
        for (int i0 = tid; i0 < snap.n_idx; i0 += nt4) {
            int32_t x[4], y[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const int i = i0 + u * (int)blockDim.x; x[u] = i < snap.n_idx ? snap.rbv[i] : 0; y[u] = i < snap.n_idx ? snap.cbv[i] : 0; }
#pragma unroll
            for (int u = 0; u < 4; u++) { const int i = i0 + u * (int)blockDim.x; if (i < snap.n_idx) { rbv[i] = x[u]; cbv[i] = y[u]; } }
        }
        for (int i0 = tid; i0 < H || i0 < s.W; i0 += nt4) {
            int32_t x[4], y[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const int i = i0 + u * (int)blockDim.x; x[u] = i < H ? snap.vibr[i] : 0; y[u] = i < s.W ? snap.vibc[i] : 0; }
#pragma unroll
            for (int u = 0; u < 4; u++) { const int i = i0 + u * (int)blockDim.x; if (i < H) vibr[i] = x[u]; if (i < s.W) vibc[i] = y[u]; }
        }
    };
    if (COW) {
        // restore() without moving a byte of the slot: the node starts from the saved root, so everything it starts from is READ
        // from there -- column 0, the cost row and the row / column maps go straight into LDS (independent loads, one trip), the
        // cut variables' rows are looked up in the root's maps.  No dirty-row scan, no row copies, no map copies: a row the
        // previous nodes wrote stays flagged (dirty: the slot differs from the root there) and is simply not current (L.cur).
        // Slot 0 is the engine's live tableau: its global maps are kept as before.
        // (every load of a turn leaves before the first LDS store: two rows and two columns per thread and turn -- as three plain loops
        //  this was up to six dependent trips for a tableau taller / wider than the workgroup)
        for (int i0 = tid; i0 < H || i0 < s.ld; i0 += 2 * (int)blockDim.x) {
            // clamped indexes, unconditional loads and stores (a clamped element gets its own value once more): no branch between the loads
            const int i1 = i0 + (int)blockDim.x;
            const int h0 = min(i0, H - 1), h1 = min(i1, H - 1), c0 = min(i0, s.ld - 1), c1 = min(i1, s.ld - 1), w0 = min(i0, s.W - 1), w1 = min(i1, s.W - 1);
            const double rh0 = snap.rhs[h0], rh1 = snap.rhs[h1];
            const int32_t vr0 = snap.vibr[h0], vr1 = snap.vibr[h1];
            const double a0 = snap.A[c0], a1 = snap.A[c1];
            const int32_t vc0 = snap.vibc[w0], vc1 = snap.vibc[w1];
            L.cur[h0] = 0; L.rhs[h0] = rh0; L.vibr[h0] = vr0;
            L.cur[h1] = 0; L.rhs[h1] = rh1; L.vibr[h1] = vr1;
            L.r0[c0] = a0;
            L.r0[c1] = a1;
            L.vibc[w0] = vc0;
            L.vibc[w1] = vc1;
        }
        if (slot == 0) restore_maps();
        if (tid == 0) {
            st->H = H;
            st->last_element_index = lei0;
            st->err = ERR_NONE;
        }
        __syncthreads();
        WL_MARK(0);
        WL_MARK(1);
        cut_err = add_cuts_waves(s, cuts, slot, node, cap_rows, snap.A, snap.rbv, snap.cbv, &L, H, lei0, cut_a, cut_n);
        __syncthreads();
        WL_MARK(2);
    } else {
    // restore(): the dirty rows, found by all threads at once and compacted into the LDS list
    if (tid == 0) sm.n_list = 0;
    __syncthreads();
    for (int r = tid; r < H; r += blockDim.x)
        if (dirty[r]) L.list[atomicAdd(&sm.n_list, 1)] = r;
    __syncthreads();
    const int n = sm.n_list;
    if (s.cnt && tid == 0) atomicAdd(s.cnt + CNT_RESTORED, (cnt_t)n);
    const double2* src = reinterpret_cast<const double2*>(snap.A);
    double2* dst = reinterpret_cast<double2*>(A);
    for (int i = w; i < n; i += nw) {
        const int r = L.list[i];
        wglds_copy_row(dst + (long long)r * ld2, src + (long long)r * ld2, ld2, lane);
        if (lane == 0) { dirty[r] = 0; rhs[r] = snap.rhs[r]; }
    }
    __syncthreads();
    WL_MARK(0);
    restore_maps();
    for (int r = tid; r < H; r += blockDim.x) L.cur[r] = 0;  // every row of the root is as saved
    if (tid == 0) {
        st->H = H;
        st->last_element_index = lei0;
        st->err = ERR_NONE;
    }
    __syncthreads();
    WL_MARK(1);
    add_cuts_waves(s, cuts, slot, node, cap_rows, nullptr, nullptr, nullptr, nullptr, -1, -1, cut_a, cut_n);
    __syncthreads();
    WL_MARK(2);
    }
    if (OPT && s.n_opt > 0 && snap.oo) {  // restore(): the optional objective rows of the saved root (backup.ts:94-104) into the slot's copy
        double* oo = s.oo + (long long)slot * s.oo_stride;
        for (long long i = THREADS == 512 ? wgl_tid() : tid; i < s.oo_stride; i += blockDim.x) oo[i] = snap.oo[i];
        __syncthreads();
    }
    const Ctx c = slot_ctx(s, slot, check_cycles);
    WgLds Ln = L;
    if (snap.AT) { Ln.snapT = snap.AT; Ln.ldT = snap.ldT; Ln.Hs = H; }
    if (COW) {
        Ln.snapA = snap.A; Ln.Hs = H; Ln.cow = true; Ln.preloaded = true;
        Ln.err_hint = cut_err;
        Ln.H_hint = cut_err == (int)ERR_NONE ? H + cut_n : H;
    }
    // (copy-on-write nodes: the error word and the height the solve ends with are known here -- no global re-reads in front of the read-back)
    const int err_fin = simplex_wg_lds<(THREADS >= 1024 ? WGL_UN1024 : WGL_UN512), OPT, (THREADS >= 1024 || WGL_PF512)>(c, sm, Ln, iters_cap);
#ifdef JSLP_DEBUG_WGLDS
    wl_prev = __builtin_amdgcn_s_memtime();
#endif
    if (COW && s.cnt) {  // rows of the root this node wrote = what a restore() before the next node has to bring back
        int n_mine = 0;
        for (int r = tid; r < H; r += blockDim.x) n_mine += L.cur[r] ? 1 : 0;
        for (int off = 32; off > 0; off >>= 1) n_mine += __shfl_down(n_mine, off, 64);
        if (lane == 0 && n_mine) atomicAdd(s.cnt + CNT_RESTORED, (cnt_t)n_mine);
    }
    if ((COW ? err_fin : (int)st->err) == ERR_NONE) gather_slot_lds(s, Ln, slot, rhs_out, rows_out, state_out, out_stride, o, COW ? Ln.H_hint : -1);
    else gather_slot(s, slot, rhs_out, rows_out, state_out, out_stride, o);
    __syncthreads();
    WL_MARK(12);
    return true;
}

// Slot 0 after a copy-on-write node: the rows earlier nodes wrote and the last one did not (dirty, not current) come back from
// the saved root.  The dirty flags are read by all threads at once and the stale rows compacted into the LDS list (a wave per row
// reading one flag per trip took ~60 dependent global loads for Monster_II's 945 rows: 70 us behind the completion flag, which
// the NEXT node of a sequential walk queues behind).
__device__ __forceinline__ void slot0_whole_again(const Slots& s, const Snapshot& snap, SmemL& sm, const WgLds& L) {
    const int H = s.st[0].s_H, ld2 = s.ld / 2, lane = threadIdx.x & 63;
    uint8_t* dirty = s.dirty;
    if (threadIdx.x == 0) sm.n_list = 0;
    __syncthreads();
    for (int r = threadIdx.x; r < H; r += blockDim.x)
        if (dirty[r] != 0 && !L.cur[r]) L.list[atomicAdd(&sm.n_list, 1)] = r;
    __syncthreads();
    const int n = sm.n_list;
    const double2* src = reinterpret_cast<const double2*>(snap.A);
    double2* dst = reinterpret_cast<double2*>(s.A);
    for (int i = threadIdx.x >> 6; i < n; i += blockDim.x >> 6) {
        const int r = L.list[i];
        wglds_copy_row(dst + (long long)r * ld2, src + (long long)r * ld2, ld2, lane);
        if (lane == 0) dirty[r] = 0;
    }
}

// COW (the single-node call of the sequential services, one workgroup on slot 0 = the engine's live tableau): the node starts from
// the saved root without restoring anything first (node_lds_run<.., COW>: what it starts from is READ from the root); the rows
// earlier nodes wrote and this one did not are brought back from the root AFTER the outcome has left and the completion flag is
// up -- the host is already walking the tree while the slot is made whole again (the next launch queues behind this one).
template <int THREADS, bool OPT = false, bool COW = false>
__global__ void __launch_bounds__(THREADS, THREADS == 512 ? JSLP_NODE512_WAVES : 4) k_node_lds(Slots s, Snapshot snap, Cuts cuts, int first_node, int check_cycles,
                                                      int iters_cap, int cap_rows, double* rhs_out, int32_t* rows_out,
                                                      DevState* state_out, int out_stride, int first_out,
                                                      unsigned* done_flag, unsigned done_seq, int* done_count = nullptr) {
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
    __shared__ SmemL sm;
    const WgLds L = wglds_carve(lds_dyn, s.ld, cap_rows);
    const bool ran = node_lds_run<THREADS, COW, OPT>(s, snap, cuts, sm, L, blockIdx.x, first_node + blockIdx.x, first_out + blockIdx.x, check_cycles,
                                                     iters_cap, cap_rows, rhs_out, rows_out, state_out, out_stride);
    if (done_flag) {
        // (round 5: a small BATCH announces itself the same way -- every workgroup makes its outcome visible system-wide and counts itself
        //  in; the one that completes the count raises the flag the host polls: no stream synchronisation, whose wake-up costs a
        //  speculative tree tens of microseconds per dependent batch)
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            bool last = true;
            if (done_count) {
                last = atomicAdd(done_count, 1) == (int)gridDim.x - 1;
                if (last) { *done_count = 0; __threadfence_system(); }  // (the next launch is ordered behind this one on the stream)
            }
            if (last) __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (COW && ran && blockIdx.x == 0) slot0_whole_again(s, snap, sm, L);
}

// A whole batch in ONE launch: as many workgroups as the chip keeps resident, each on its own slot, each pulling the next node
// from a queue (an atomic counter) until the batch is empty -- no group boundaries, hence no idle tail per group; `order`
// hands the nodes out most-cuts-first (the cut count predicts the repair pivots: longest-processing-time-first keeps the
// last workgroups to finish on the cheap nodes).  Node k's outcome goes to index k whatever workgroup / slot evaluated it.
// OPT (round 4): models with optional objectives -- soft-constraint MILPs -- through the queue too (eager restores: every node takes the
// saved root's objective rows into its slot's copy, backup.ts:94-104; a build of its own like the other OPT kernels)
template <int THREADS, bool COW, bool OPT = false>
__global__ void __launch_bounds__(THREADS, THREADS == 512 ? JSLP_NODE512_WAVES : 4) k_node_queue(Slots s, Snapshot snap, Cuts cuts, int n_nodes, const int32_t* order,
                                                      int* queue, int check_cycles, int iters_cap, int cap_rows, double* rhs_out,
                                                      int32_t* rows_out, DevState* state_out, int out_stride) {
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
    __shared__ SmemL sm;
    __shared__ int q_next;
    const WgLds L = wglds_carve(lds_dyn, s.ld, cap_rows);
    bool ran = false;
    for (;;) {
        if (threadIdx.x == 0) q_next = atomicAdd(queue, 1);
        __syncthreads();
        const int k = q_next;
        __syncthreads();
        if (k >= n_nodes) break;
        ran = true;
        const int node = order ? order[k] : k;
        int slot = blockIdx.x;
        asm volatile("" : "+s"(slot));  // opaque per iteration: nothing derived from the slot is hoisted and kept live across nodes
        node_lds_run<THREADS, COW, OPT>(s, snap, cuts, sm, L, slot, node, node, check_cycles, iters_cap, cap_rows, rhs_out, rows_out, state_out, out_stride);
        __syncthreads();
    }
    // slot 0 is also the engine's live tableau: leave it whole (the last node this workgroup evaluated), as the other batch shapes
    // do.  The other slots keep their stale rows (flagged dirty) until someone restores them.
    if (COW && blockIdx.x == 0 && ran) slot0_whole_again(s, snap, sm, L);
}
