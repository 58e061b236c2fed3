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
// Those functions live in jslp_core.inc.h, written over `real_t` and compiled below for double (the engine) and for
// float (namespace f32: the fp32 twin used by the fp32-vs-fp64 sweep only).  This file adds what exists once:
// snapshots / checkpoints / cuts / MIR cuts / read-back, the fused one-launch-per-pivot phase 2 and the
// register-resident whole-solve kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define JSLP_WG_THREADS 1024      // workgroup size of the select step / the per-node simplex kernel
#define JSLP_UPD_THREADS 256      // workgroup size of the streaming row update
#define JSLP_UPD_ROWS 8           // rows per workgroup in the streaming row update
#define JSLP_UPD_COLS (JSLP_UPD_THREADS * 2)

enum { ST_RUNNING = 0, ST_DONE = 1, ST_PHASE1_DONE = 2, ST_P1_SLOW = 3 };  // ST_P1_SLOW: the fused phase 1 hands ONE pivot to k_select + k_update (see k_fused_p1)
enum { ERR_NONE = 0, ERR_HIST_FULL = 1, ERR_ITER_LIMIT = 2, ERR_CUT_ARG = 3, ERR_CAPACITY = 4, ERR_BARRIER = 5, ERR_NOT_SYNCED = 6 };

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
    // fused phase-2 pipeline
    int32_t gen;          // snapshot generation (>= 1) this slot's rows are in sync with except its dirty rows; 0 = unknown
    int32_t s_gen;        // slot 0 only: generation of the current snapshot
    int32_t f_pc;         // entering column chosen for the NEXT launch's pivot
    int32_t f_final_buf;  // which ping-pong buffer holds the tableau when the pipeline stopped
    int32_t mir_added;    // rows appended by the last k_mir_cuts
    int32_t rhs_valid;    // the slot's contiguous RHS mirror (Ctx::rhs) equals column 0 (kept by the per-node kernel only)
};
// Work counters (jslp_work_counters), one block per engine, advanced with device-scope atomics only while counting is on
enum { CNT_CELLS = 0, CNT_ROWS = 1, CNT_RESTORED = 2, CNT_N = 4, CNT_DBG = 8, CNT_ALLOC = 40 };  // CNT_DBG..: cycle accumulators of debug builds
typedef unsigned long long cnt_t;

// ---- the simplex core, compiled for both scalar types (see jslp_core.inc.h) -------------------------------------
__device__ __forceinline__ double rmul_rn(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double rsub_rn(double a, double b) { return __dsub_rn(a, b); }
__device__ __forceinline__ float rmul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float rsub_rn(float a, float b) { return __fsub_rn(a, b); }
namespace f64 {  // (a namespace of its own so that argument-dependent lookup from f32 cannot reach it)
typedef double real_t;
typedef double2 real2_t;
#include "jslp_core.inc.h"
}  // namespace f64
using namespace f64;
namespace f32 {
typedef float real_t;
typedef float2 real2_t;
#include "jslp_core.inc.h"
}  // namespace f32


struct SnapshotW {
    double* A;
    int32_t *vibr, *vibc, *rbv, *cbv;
    int32_t n_idx;
    double* oo;
    double* rhs;  // contiguous copy of column 0
};
// restore (backup.ts:53-105): snapshot -> slots [first_slot, first_slot+gridDim.y).  Grid-stride copy.
struct Snapshot {
    const double* A;
    const int32_t *vibr, *vibc, *rbv, *cbv;
    int32_t n_idx;
    const double* oo;  // n_opt * ld; nullptr = leave the optional objectives alone (restoreCheckpoint does)
    // H < 0: the saved root (scalars in slot 0's DevState).  Otherwise a checkpoint (incremental-branch-and-cut.ts:72-107)
    // with its own scalars; the dirty-row shortcut does not apply to it.
    int32_t H, last_element_index;
    const double* rhs;  // contiguous copy of column 0
    const double* AT;   // the saved root transposed (column stride ldT), or nullptr: see WgLds::snapT
    int32_t ldT;
};
// snapshot -> its transpose (32 x 32 tiles through LDS, both sides coalesced); rows = the saved root's height
__global__ void __launch_bounds__(256) k_snapshot_transpose(const DevState* st0, const double* A, int ld, int W, double* AT, int ldT) {
    __shared__ double tile[32][33];
    const int H = st0->s_H;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    if (r0 >= H) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < H && c < W) ? A[(long long)r * ld + c] : 0.0;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (c < W && r < H) AT[(long long)c * ldT + r] = tile[tx][j];
    }
}
__global__ void __launch_bounds__(256) k_restore(Slots s, Snapshot snap, int first_slot) {
    const int slot = first_slot + blockIdx.y;
    DevState* st = s.st + slot;
    const bool root = snap.H < 0;
    const int H = root ? s.st[0].s_H : snap.H;  // every slot shares slot 0's snapshot scalars
    const int gen = s.st[0].s_gen;
    const bool incremental = root && gen != 0 && st->gen == gen;  // this slot already holds the snapshot except for its dirty rows
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nt = (long long)gridDim.x * blockDim.x;
    const double2* src = reinterpret_cast<const double2*>(snap.A);
    double2* dst = reinterpret_cast<double2*>(s.A + (long long)slot * s.A_stride);
    uint8_t* dirty = s.dirty + (long long)slot * s.pcol_stride;
    double* rhs = s.rhs + (long long)slot * s.pcol_stride;
    if (!incremental) {
        const long long n2 = (long long)H * s.ld / 2;
        for (long long i = tid; i < n2; i += nt) dst[i] = src[i];
        for (long long i = tid; i < s.pcol_stride; i += nt) dirty[i] = 0;
        for (long long i = tid; i < H; i += nt) rhs[i] = snap.rhs[i];
        if (s.cnt && tid == 0) atomicAdd(s.cnt + CNT_RESTORED, (cnt_t)H);
    } else {
        // one wave per dirty row: rows are found by a strided scan of the byte flags
        const int lane = threadIdx.x & 63;
        const long long wave = tid >> 6, nwaves = nt >> 6;
        const int ld2 = s.ld / 2;
        for (long long r = wave; r < H; r += nwaves) {
            if (!dirty[r]) continue;
            const double2* a = src + r * ld2;
            double2* b = dst + r * ld2;
            for (int i = lane; i < ld2; i += 64) b[i] = a[i];
            if (lane == 0) { dirty[r] = 0; rhs[r] = snap.rhs[r]; if (s.cnt) atomicAdd(s.cnt + CNT_RESTORED, (cnt_t)1); }
        }
    }
    int32_t* vibr = s.vibr + (long long)slot * s.vibr_stride;
    for (long long i = tid; i < H; i += nt) vibr[i] = snap.vibr[i];
    int32_t* vibc = s.vibc + (long long)slot * s.vibc_stride;
    for (long long i = tid; i < s.W; i += nt) vibc[i] = snap.vibc[i];
    int32_t* rbv = s.rbv + (long long)slot * s.idx_stride;
    int32_t* cbv = s.cbv + (long long)slot * s.idx_stride;
    for (long long i = tid; i < snap.n_idx; i += nt) { rbv[i] = snap.rbv[i]; cbv[i] = snap.cbv[i]; }
    if (s.n_opt > 0 && snap.oo) {  // backup.ts:94-104
        double* oo = s.oo + (long long)slot * s.oo_stride;
        for (long long i = tid; i < s.oo_stride; i += nt) oo[i] = snap.oo[i];
    }
    if (tid == 0) {
        st->H = H;
        st->last_element_index = root ? s.st[0].s_last_element_index : snap.last_element_index;
        st->err = ERR_NONE;
        if (!incremental) st->rhs_valid = 1;  // (an incremental restore keeps whatever the mirror's state was)
    }
}
// second half of restore(): record that the slots are in sync with the saved root (a separate tiny launch: k_restore's
// workgroups all read st->gen, so none of them may write it); after a checkpoint restore they are in sync with nothing
__global__ void k_restore_commit(Slots s, int first_slot, int from_root) {
    s.st[first_slot + blockIdx.x].gen = from_root ? s.st[0].s_gen : 0;
}

// Math.max(0, x) / Math.min(0, x) with JavaScript's treatment of NaN and signed zeros
__device__ __forceinline__ double js_max0(double x) { return x != x ? x : (x > 0 ? x : 0.0); }
__device__ __forceinline__ double js_min0(double x) {
    return x != x ? x : ((x < 0 || (x == 0 && (__double_as_longlong(x) < 0))) ? x : 0.0);
}

// applyMIRCuts (cutting-strategies.ts:199-212) on slot 0: rows 1..H-1 are scanned in order and the first <= max_cuts rows
// whose basic variable is an integer variable with a fractional value each append one lower-bound MIR cut
// (addLowerBoundMIRCut, :74-135).  One workgroup: the scan is an ordered compaction (ballot + popcount), the new rows are
// elementwise in the columns.
__global__ void __launch_bounds__(256) k_mir_cuts(Slots s, const uint8_t* is_int, int max_cuts, int cap_rows) {
    __shared__ int sel[64];
    __shared__ unsigned long long masks[4];
    DevState* st = s.st;
    double* A = s.A;
    int32_t* vibr = s.vibr;
    const int32_t* vibc = s.vibc;
    const int H = st->H, W = s.W, ld = s.ld;
    const double precision = s.precision;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = 0;
    for (int r0 = 1; r0 < H && base < max_cuts; r0 += 256) {
        const int r = r0 + threadIdx.x;
        bool ok = false;
        if (r < H) {
            const int v = vibr[r];
            if (v >= 0 && is_int[v]) {  // :82-85
                const double rhs = A[(long long)r * ld];
                const double f = rhs - floor(rhs);
                ok = !(f < precision || f > 1 - precision);  // :88-90
            }
        }
        const unsigned long long m = __ballot(ok);
        if (lane == 0) masks[wave] = m;
        __syncthreads();
        int before = __popcll(m & ((1ull << lane) - 1)), total = 0;
        for (int k = 0; k < 4; k++) {
            const int n = __popcll(masks[k]);
            if (k < wave) before += n;
            total += n;
        }
        if (ok && base + before < max_cuts) sel[base + before] = r;
        base += total;
        __syncthreads();
    }
    const int n = base < max_cuts ? base : max_cuts;
    if (H + n > cap_rows) {
        if (threadIdx.x == 0) { st->err = ERR_CAPACITY; st->mir_added = 0; }
        return;
    }
    for (int k = 0; k < n; k++) {
        const double* src = A + (long long)sel[k] * ld;
        double* dst = A + (long long)(H + k) * ld;
        const double rhs = src[0];
        const double f = rhs - floor(rhs);
        for (int c = threadIdx.x; c < ld; c += 256) {
            double out = 0.0;
            if (c == 0) {
                out = floor(rhs) - rhs;  // :112 then :128-130
            } else if (c < W) {
                const double a = src[c];
                const int v = vibc[c];
                double cut;
                if (v >= 0 && is_int[v]) {  // :117-123
                    const double fl = floor(a);
                    cut = fl + js_max0(a - fl - f) / (1 - f);
                } else {
                    cut = js_min0(a / (1 - f));  // :124-125
                }
                out = cut - a;  // :128-130
            }
            dst[c] = out;
            if (c == 0) s.rhs[H + k] = out;
        }
    }
    if (threadIdx.x == 0) {
        for (int k = 0; k < n; k++) {  // :104-109
            const int slack = st->last_element_index++;
            vibr[H + k] = slack;
            s.rbv[slack] = H + k;
            s.cbv[slack] = -1;
            s.dirty[H + k] = 1;
        }
        st->H = H + n;
        st->mir_added = n;
    }
}

// createCheckpoint (incremental-branch-and-cut.ts:55-70): slot 0 -> a checkpoint buffer; the snapshot generation and the
// saved root are not touched
__global__ void __launch_bounds__(256) k_checkpoint(Slots s, SnapshotW ck, int H) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nt = (long long)gridDim.x * blockDim.x;
    const long long n2 = (long long)H * s.ld / 2;
    const double2* src = reinterpret_cast<const double2*>(s.A);
    double2* dst = reinterpret_cast<double2*>(ck.A);
    for (long long i = tid; i < n2; i += nt) dst[i] = src[i];
    for (long long i = tid; i < H; i += nt) ck.vibr[i] = s.vibr[i];
    for (long long i = tid; i < s.W; i += nt) ck.vibc[i] = s.vibc[i];
    for (long long i = tid; i < ck.n_idx; i += nt) { ck.rbv[i] = s.rbv[i]; ck.cbv[i] = s.cbv[i]; }
    for (long long i = tid; i < H; i += nt) ck.rhs[i] = s.A[i * s.ld];
}

// save (backup.ts:13-51): slot 0 -> snapshot
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
    for (long long i = tid; i < H; i += nt) snap.rhs[i] = s.A[i * s.ld];
    if (s.n_opt > 0)  // backup.ts:37-43
        for (long long i = tid; i < s.oo_stride; i += nt) snap.oo[i] = s.oo[i];
    if (tid == 0) {
        st->s_H = H;
        st->s_last_element_index = st->last_element_index;
        st->s_gen += 1;   // every slot's copy is stale now ...
        st->gen = 0;      // ... including slot 0 (it equals the snapshot, but its dirty flags are not maintained by every path)
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
__device__ __forceinline__ void add_cuts_slot(const Slots& s, const Cuts& cuts, int slot, int node, int cap_rows) {
    DevState* st = s.st + slot;
    double* A = s.A + (long long)slot * s.A_stride;
    double* rhs = s.rhs + (long long)slot * s.pcol_stride;
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
                if (col == 0) { v = sign * value; rhs[H + h] = v; }
                else if (col == var_col) v = sign;
                cut[col] = v;
            }
        } else {  // basic variable: negated copy of its row (:54-62)
            const double* src = A + (long long)var_row * ld;
            for (int col = threadIdx.x; col < ld; col += blockDim.x) {
                double v = 0.0;
                if (col == 0) { v = sign * (value - src[0]); rhs[H + h] = v; }
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
__global__ void __launch_bounds__(256) k_add_cuts(Slots s, Cuts cuts, int first_slot, int first_node, int cap_rows) {
    add_cuts_slot(s, cuts, first_slot + blockIdx.x, first_node + blockIdx.x, cap_rows);
}

// Read-back for the host tree: RHS column + varIndexByRow of each slot, and the slot's state.
// out_stride < 0: compact read-back (jslp_engine_relax_watched) -- per watched variable its row (rowByVarIndex) and its RHS
// cell, -out_stride entries per node
__device__ __forceinline__ void gather_slot(const Slots& s, int slot, double* rhs, int32_t* rows, DevState* states,
                                            int out_stride, int o) {
    const DevState* st = s.st + slot;
    const double* A = s.A + (long long)slot * s.A_stride;
    const double* mirror = s.rhs + (long long)slot * s.pcol_stride;
    const bool mirrored = st->rhs_valid != 0;
    const int32_t* vibr = s.vibr + (long long)slot * s.vibr_stride;
    const int H = st->H;
    if (out_stride < 0) {
        const int32_t* rbv = s.rbv + (long long)slot * s.idx_stride;
        const int n = -out_stride;
        for (int i = threadIdx.x; i < n && i < s.n_watch; i += blockDim.x) {
            const int r = rbv[s.watch[i]];
            const bool basic = r > 0 && r < H;
            if (rows) rows[(long long)o * n + i] = basic ? r : -1;
            if (rhs) rhs[(long long)o * n + i] = basic ? (mirrored ? mirror[r] : A[(long long)r * s.ld]) : 0.0;
        }
        if (threadIdx.x == 0) states[o] = *st;
        return;
    }
    for (int r = threadIdx.x; r < H; r += blockDim.x) {
        if (rhs) rhs[(long long)o * out_stride + r] = mirrored ? mirror[r] : A[(long long)r * s.ld];
        if (rows) rows[(long long)o * out_stride + r] = vibr[r];
    }
    if (threadIdx.x == 0) states[o] = *st;
}
__global__ void __launch_bounds__(256) k_gather(Slots s, int first_slot, double* rhs, int32_t* rows, DevState* states,
                                                int out_stride, int first_out) {
    gather_slot(s, first_slot + blockIdx.x, rhs, rows, states, out_stride, first_out + blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------
// ONE branch-and-bound child in ONE launch (slot 0, from the saved root): the restore of the rows the previous node
// dirtied, the commit, addCutConstraints, the whole simplex() and the read-back, which goes straight into the pinned
// host buffer (and the cut list is read straight from one) -- a sequential tree walk pays one launch and one stream
// synchronisation per node instead of five launches and two copies.  Only valid when slot 0 is in sync with the current
// snapshot generation (the host tracks that and otherwise uses k_restore / k_add_cuts / k_simplex_wg / k_gather once).
// ---------------------------------------------------------------------------------------------------
// Workgroup blockIdx.x evaluates node (first_node + blockIdx.x) on slot blockIdx.x and writes outcome (first_out +
// blockIdx.x).  <1024, 4096> for one node (outputs may point into pinned host memory), <512, 2048> for batches.
// (second launch bound = waves per SIMD to fit: 512-thread workgroups are meant to run FOUR per CU, i.e. <= 64 VGPRs)
template <int THREADS, int CAP>
__global__ void __launch_bounds__(THREADS, THREADS == 512 ? 8 : 4) k_node_wg(Slots s, Snapshot snap, Cuts cuts, int first_node, int check_cycles,
                                                     int iters_cap, int cap_rows, double* rhs_out, int32_t* rows_out,
                                                     DevState* state_out, int out_stride, int first_out,
                                                     unsigned* done_flag, unsigned done_seq) {
    __shared__ Smem sm;
    __shared__ ActSmem<CAP> act;
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
    // restore(): the dirty rows, found by all threads at once and compacted into the LDS list the update uses later
    if (tid == 0) act.n = 0;
    __syncthreads();
    for (int r = tid; r < H; r += blockDim.x)
        if (dirty[r]) {
            const int idx = atomicAdd(&act.n, 1);
            if (idx < CAP) act.row[idx] = r;
        }
    __syncthreads();
    const int n = act.n;
    if (s.cnt && tid == 0) atomicAdd(s.cnt + CNT_RESTORED, (cnt_t)n);
    const double2* src = reinterpret_cast<const double2*>(snap.A);
    double2* dst = reinterpret_cast<double2*>(A);
    if (n <= CAP) {
        for (int i = w; i < n; i += nw) {
            const int r = act.row[i];
            for (int k = lane; k < ld2; k += 64) dst[(long long)r * ld2 + k] = src[(long long)r * ld2 + k];
            if (lane == 0) { dirty[r] = 0; rhs[r] = snap.rhs[r]; }
        }
    } else {
        for (int r = w; r < H; r += nw) {
            if (!dirty[r]) continue;
            for (int k = lane; k < ld2; k += 64) dst[(long long)r * ld2 + k] = src[(long long)r * ld2 + k];
            if (lane == 0) { dirty[r] = 0; rhs[r] = snap.rhs[r]; }
        }
    }
    int32_t* vibr = s.vibr + (long long)slot * s.vibr_stride;
    int32_t* vibc = s.vibc + (long long)slot * s.vibc_stride;
    int32_t* rbv = s.rbv + (long long)slot * s.idx_stride;
    int32_t* cbv = s.cbv + (long long)slot * s.idx_stride;
    for (int i = tid; i < H; i += blockDim.x) vibr[i] = snap.vibr[i];
    for (int i = tid; i < s.W; i += blockDim.x) vibc[i] = snap.vibc[i];
    for (int i = tid; i < snap.n_idx; i += blockDim.x) { rbv[i] = snap.rbv[i]; cbv[i] = snap.cbv[i]; }
    if (s.n_opt > 0 && snap.oo) {
        double* oo = s.oo + (long long)slot * s.oo_stride;
        for (long long i = tid; i < s.oo_stride; i += blockDim.x) oo[i] = snap.oo[i];
    }
    if (tid == 0) {
        st->H = H;
        st->last_element_index = s.st[0].s_last_element_index;
        st->err = ERR_NONE;
    }
    __syncthreads();
    add_cuts_slot(s, cuts, slot, node, cap_rows);
    __syncthreads();
    const Ctx c = slot_ctx(s, slot, check_cycles);
    simplex_wg(c, sm, act, iters_cap);
    __syncthreads();
    gather_slot(s, slot, rhs_out, rows_out, state_out, out_stride, o);
    if (done_flag) {
        // one node, outcome written straight into pinned host memory: the host polls this flag instead of paying a stream
        // synchronisation (every thread makes its own writes visible system-wide, then ONE release store of the sequence number)
        __threadfence_system();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- upload (tableau.ts:292-380 hand-over): the matrix arrives by DMA; ONE blob brings [vibr | vibc | unrestricted list]
// and this kernel builds the inverse maps, the flags and the state on the device (one workgroup) -----------------------------
struct UploadBlob {
    const int32_t* vibr;  // H entries
    const int32_t* vibc;  // W entries
    const int32_t* unr;   // n_unr variable indexes
    int32_t H, n_unr, n_idx, cap_rows;
};
__global__ void __launch_bounds__(1024) k_upload_finish(Slots s, UploadBlob b, uint8_t* unr_flags, uint8_t* isint_flags) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < b.n_idx; i += nt) { s.rbv[i] = -1; s.cbv[i] = -1; unr_flags[i] = 0; isint_flags[i] = 0; }
    for (int i = tid; i < b.cap_rows; i += nt) { s.vibr[i] = i < b.H && i > 0 ? b.vibr[i] : -1; s.dirty[i] = 0; }
    for (int i = tid; i < s.W; i += nt) s.vibc[i] = i > 0 ? b.vibc[i] : -1;
    __syncthreads();
    for (int r = 1 + tid; r < b.H; r += nt) s.rbv[b.vibr[r]] = r;
    for (int c = 1 + tid; c < s.W; c += nt) s.cbv[b.vibc[c]] = c;
    for (int i = tid; i < b.n_unr; i += nt) unr_flags[b.unr[i]] = 1;
    if (tid == 0) {
        DevState st;
        memset(&st, 0, sizeof st);
        st.H = b.H;
        st.last_element_index = s.W + b.H - 2;  // tableau.ts:312-316
        st.status = ST_DONE;
        st.phase = 1;
        st.feasible = 1;
        st.bounded = 1;
        st.unbounded_var = -1;
        *s.st = st;
    }
}
// non-zero cells of the uploaded tableau (the host picks the launch shape by density: see use_wg_single); *out zeroed by the host
__global__ void __launch_bounds__(256) k_count_nnz(const double* A, int H, int ld, unsigned long long* out) {
    const long long n = (long long)H * ld;
    unsigned long long mine = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        mine += A[i] != 0.0 ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(out, mine);
}

// strided host layout (row stride W) -> device layout (row stride ld) for a tableau that arrived as ONE contiguous DMA
__global__ void __launch_bounds__(256) k_repack(double* A, const double* packed, int H, int W, int ld) {
    const long long n = (long long)H * ld;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / ld), c = (int)(i - (long long)r * ld);
        A[i] = c < W ? packed[(long long)r * W + c] : 0.0;
    }
}

// ---- resident kernel safety net: slot 0 (matrix rows [0, H), maps, state) <-> a backup taken before the cooperative launch
__global__ void __launch_bounds__(256) k_res_backup(Slots s, SnapshotW bk, DevState* st_bk, int H, int to_backup) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nt = (long long)gridDim.x * blockDim.x;
    const long long n2 = (long long)H * s.ld / 2;
    double2* live = reinterpret_cast<double2*>(s.A);
    double2* back = reinterpret_cast<double2*>(bk.A);
    if (to_backup) {
        for (long long i = tid; i < n2; i += nt) back[i] = live[i];
        for (long long i = tid; i < H; i += nt) bk.vibr[i] = s.vibr[i];
        for (long long i = tid; i < s.W; i += nt) bk.vibc[i] = s.vibc[i];
        for (long long i = tid; i < bk.n_idx; i += nt) { bk.rbv[i] = s.rbv[i]; bk.cbv[i] = s.cbv[i]; }
        if (tid == 0) *st_bk = *s.st;
    } else {
        for (long long i = tid; i < n2; i += nt) live[i] = back[i];
        for (long long i = tid; i < H; i += nt) s.vibr[i] = bk.vibr[i];
        for (long long i = tid; i < s.W; i += nt) s.vibc[i] = bk.vibc[i];
        for (long long i = tid; i < bk.n_idx; i += nt) { s.rbv[i] = bk.rbv[i]; s.cbv[i] = bk.cbv[i]; }
        if (tid == 0) *s.st = *st_bk;
    }
}

// ---- device pool: a member that just received the primary's saved root by peer copy adopts it as ITS saved root ------------
__global__ void k_adopt_root(Slots s, int s_H, int s_last_element_index) {
    DevState* st = s.st;
    st->s_H = s_H;
    st->s_last_element_index = s_last_element_index;
    st->s_gen += 1;
    st->gen = 0;
    st->H = s_H;
    st->last_element_index = s_last_element_index;
    st->status = ST_DONE;
    st->phase = 2;
    st->feasible = 1;
    st->bounded = 1;
    st->unbounded_var = -1;
    st->err = ERR_NONE;
    st->rhs_valid = 0;
}

#include "jslp_wglds.hip.h"

// ---- fp32 twin (jslp_engine_simplex_f32): narrow the live fp64 tableau into an fp32 slot / widen the read-back ----------
__global__ void __launch_bounds__(256) k32_convert(Slots s, f32::Slots d, int iters_unused) {
    const DevState* st = s.st;
    const int H = st->H;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nt = (long long)gridDim.x * blockDim.x;
    const long long n = (long long)H * s.ld;
    for (long long i = tid; i < n; i += nt) d.A[i] = (float)s.A[i];
    for (long long i = tid; i < H; i += nt) d.vibr[i] = s.vibr[i];
    for (long long i = tid; i < s.W; i += nt) d.vibc[i] = s.vibc[i];
    for (long long i = tid; i < s.idx_stride; i += nt) { d.rbv[i] = s.rbv[i]; d.cbv[i] = s.cbv[i]; }
    if (tid == 0) {
        DevState c = *st;
        c.gen = 0;
        c.trace_n = 0;
        *d.st = c;
    }
}
__global__ void __launch_bounds__(256) k32_gather(f32::Slots s, double* rhs, int32_t* rows, DevState* states) {
    const DevState* st = s.st;
    const int H = st->H;
    for (int r = threadIdx.x; r < H; r += blockDim.x) {
        if (rhs) rhs[r] = (double)s.A[(long long)r * s.ld];
        if (rows) rows[r] = s.vibr[r];
    }
    if (threadIdx.x == 0) states[0] = *st;
}

// ---- the two large-LP kernels live in their own files ------------------------------------------------------------
#include "jslp_fused.hip.h"
#include "jslp_resident.hip.h"
