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

enum { ST_RUNNING = 0, ST_DONE = 1, ST_PHASE1_DONE = 2 };
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
};
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
            if (lane == 0) { dirty[r] = 0; rhs[r] = snap.rhs[r]; }
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
__device__ __forceinline__ void gather_slot(const Slots& s, int slot, double* rhs, int32_t* rows, DevState* states,
                                            int out_stride, int o) {
    const DevState* st = s.st + slot;
    const double* A = s.A + (long long)slot * s.A_stride;
    const double* mirror = s.rhs + (long long)slot * s.pcol_stride;
    const bool mirrored = st->rhs_valid != 0;
    const int32_t* vibr = s.vibr + (long long)slot * s.vibr_stride;
    const int H = st->H;
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
                                                     DevState* state_out, int out_stride, int first_out) {
    __shared__ Smem sm;
    __shared__ ActSmem<CAP> act;
    const int slot = blockIdx.x, node = first_node + blockIdx.x, o = first_out + blockIdx.x;
    DevState* st = s.st + slot;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
    const int gen = s.st[0].s_gen, H = s.st[0].s_H, ld2 = s.ld / 2;  // every slot shares slot 0's snapshot scalars
    if (gen == 0 || st->gen != gen) {  // must not happen (host bookkeeping): refuse rather than restore wrongly
        if (tid == 0) { st->err = ERR_NOT_SYNCED; st->status = ST_DONE; state_out[o] = *st; }
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
}

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
// Preconditions checked by the host: phase 2, no unrestricted variables, ld <= 2048, H <= 64 * 256,
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
};

__device__ __forceinline__ void fcand_consider(FCand& best, int r, double colv, double rhs, double precision) {
    // one row of the ratio test (simplex.ts:276-296), rows visited in ascending order
    if (-precision < colv && colv < precision) return;
    if (colv > 0 && precision > rhs && rhs > -precision) {
        if (r < best.rdeg) { best.rdeg = r; best.kdeg = colv; }
        return;
    }
    const double quo = rhs / colv;  // isReducedCostNegative is false without unrestricted variables
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
    double col[JSLP_F_RG];
    double rhs[JSLP_F_RG];
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

// pricing of a cost-row pair held in registers (columns c0, c0+1): simplex.ts:118-219 without unrestricted vars
__device__ __forceinline__ int price_row(double x0, double x1, int c0, const Ctx& c, Smem& sm, unsigned long long* dbg = nullptr) {
    // the lane's two columns, written out field by field (no struct select: a `Cand e = cand` inside an unrolled
    // loop was observed to keep the FIRST column's index with the SECOND column's value on gfx950 / ROCm 7.2)
    const int col1 = c0 + 1;
    const bool ok0 = c0 >= 1 && c0 < c.W && x0 > c.precision;
    const bool ok1 = col1 < c.W && x1 > c.precision;   // col1 >= 1 always
    const int b0 = c.use_partial ? (c0 - 1) / c.batch : 0;
    const int b1 = c.use_partial ? (col1 - 1) / c.batch : 0;
    double bv = c.precision;
    int bi = 0, bb = 0;
    if (ok0) { bv = x0; bi = c0; bb = b0; }
    // column c0+1 replaces column c0 only when strictly better in (batch asc, value desc); ties keep c0
    const bool take1 = ok1 && (bi == 0 || b1 < bb || (b1 == bb && x1 > bv));
    bv = take1 ? x1 : bv;
    bi = take1 ? col1 : bi;
    bb = take1 ? b1 : bb;
    Cand e;
    e.v = bv; e.i = bi; e.b = bb;
    if (dbg) {
        dbg[threadIdx.x * 4] = (unsigned long long)__double_as_longlong(e.v);
        dbg[threadIdx.x * 4 + 1] = (unsigned long long)(unsigned)e.i | ((unsigned long long)(unsigned)e.b << 32);
        dbg[threadIdx.x * 4 + 2] = (unsigned long long)__double_as_longlong(x0);
        dbg[threadIdx.x * 4 + 3] = (unsigned long long)__double_as_longlong(x1);
    }
    e = block_reduce(e, PriceFirst(), sm);
    if (dbg && threadIdx.x == 0) dbg[4096] = (unsigned long long)(unsigned)e.i;
    return e.i;
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
    const int c0 = tid * 2;
    const bool colok = c0 < ld;
    const int r_begin = b * f.rpb, r_end = min(H, r_begin + f.rpb);
    // Everything whose address does not depend on the selection is requested FIRST, before any barrier: the
    // workgroup's first row group (8 rows x 16 B per lane), its pivot-column entries, the cost row, the
    // candidates and the state.  The selection below then runs while these loads are in flight.
    double2 a[JSLP_F_RG];
    double k[JSLP_F_RG];
    double2 row0 = make_double2(0, 0);
    double k0 = 0.0;
    FCand mine = fcand_none();
    // (vmcnt retires in order: the selection's own inputs go first so that waiting for them leaves the bulk
    // row loads in flight)
    const int status = sin->status;
    const int pc_in = sin->f_pc;
    const int iters_left_in = sin->iters_left;
    if (!init) {
        if (tid < f.G) mine = cin[tid];
        k0 = pin[0];
        if (colok) row0 = *reinterpret_cast<const double2*>(Min + c0);
#pragma unroll
        for (int i = 0; i < JSLP_F_RG; i++) {
            const int r = r_begin + i;
            k[i] = r < r_end ? pin[r] : 0.0;
            a[i] = make_double2(0, 0);
            if (r < r_end && colok) a[i] = ld_stream(Min + (long long)r * ld + c0, f.nt);
        }
    }
    const bool live = init ? (status == ST_PHASE1_DONE) : (status == ST_RUNNING);
    if (!live) {  // the solve already ended (or never reached phase 2): carry the state forward
        if (b == 0 && tid == 0) { copy_state(sout, sin); if (init) sout->f_final_buf = 0; }
        return;
    }
    if (init) {
        // first phase-2 step: price the current cost row, then scan the entering column for candidates
        double2 r0 = make_double2(0, 0);
        if (colok) r0 = *reinterpret_cast<const double2*>(Min + c0);
        const int cn = price_row(r0.x, r0.y, c0, c, sm.red);
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
                if (r >= 1) fcand_consider(best, r, colv, Min[(long long)r * ld], precision);
            }
            // lanes hold rows in interleaved order: the reduction's (q, r) / min-rdeg orders are total
            best = fcand_wave_reduce(best);
            if (tid == 0) cout[b] = best;
        }
        if (b == 0 && tid == 0) {
            copy_state(sout, sin); DevState& s = *sout;
            s.status = ST_RUNNING; s.f_pc = cn; s.f_final_buf = 0; s.do_pivot = 0;
        }
        return;
    }

    // ---- STEP: pivot t ---------------------------------------------------------------------------
    const int pc = pc_in;
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
    // normalised pivot row in registers (simplex.ts:352-364; anyrow is true in phase 2, see header)
    double2 p = make_double2(0, 0);
    if (colok) {
        const double2 pv = *reinterpret_cast<const double2*>(Min + (long long)pr * ld + c0);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int col = c0 + j;
            const double val = j ? pv.y : pv.x;
            double v = 0.0;
            if (col < W) {
                const bool innz = nonzero16(val);
                v = innz ? val / quot : 0.0;
                if (col == pc) v = 1.0 / quot;
                if (innz && !nonzero16(v) && v != 0.0) v = 0.0;
            }
            if (j) p.y = v; else p.x = v;
        }
    }
    const bool v0 = nonzero16(p.x), v1 = nonzero16(p.y);
    const bool has_pc = colok && ((pc == c0) || (pc == c0 + 1));
    // updated cost row (every workgroup derives the same values), priced for pivot t+1
    double2 n0 = row0;
    if (nonzero16(k0)) {
        if (v0) n0.x = eliminate(n0.x, k0, p.x);
        if (v1) n0.y = eliminate(n0.y, k0, p.y);
        if (has_pc) { const double nv = -k0 / quot; if (pc == c0) n0.x = nv; else n0.y = nv; }
    }
    const int cn = price_row(n0.x, n0.y, c0, c, sm.red);  // 0 => optimal after this pivot

    // ---- stream my rows ---------------------------------------------------------------------------
    FCand best = fcand_none();  // kept by lanes 0..7 of wave 0: lane i sees rows r_begin+i, +8, ... in order
    for (int g0 = r_begin; g0 < r_end; g0 += JSLP_F_RG) {
        if (g0 != r_begin) {  // the first group was prefetched at the top
#pragma unroll
            for (int i = 0; i < JSLP_F_RG; i++) {
                const int r = g0 + i;
                k[i] = r < r_end ? pin[r] : 0.0;
                a[i] = make_double2(0, 0);
                if (r < r_end && colok) a[i] = ld_stream(Min + (long long)r * ld + c0, f.nt);
            }
        }
#pragma unroll
        for (int i = 0; i < JSLP_F_RG; i++) {
            const int r = g0 + i;
            if (r >= r_end) break;
            double2 x = a[i];
            if (r == pr) {
                x = p;
            } else if (nonzero16(k[i])) {
                if (v0) x.x = eliminate(x.x, k[i], p.x);
                if (v1) x.y = eliminate(x.y, k[i], p.y);
                if (has_pc) { const double nv = -k[i] / quot; if (pc == c0) x.x = nv; else x.y = nv; }
            }
            if (colok) st_stream(Mout + (long long)r * ld + c0, x, f.nt);
            if (cn != 0) {
                if (cn == c0) sm.col[i] = x.x; else if (cn == c0 + 1) sm.col[i] = x.y;
                if (tid == 0) sm.rhs[i] = x.x;
            }
        }
        if (cn != 0) {
            __syncthreads();
            if (tid < JSLP_F_RG && g0 + tid < r_end) {
                const int r = g0 + tid;
                const double colv = sm.col[tid];
                pout[r] = colv;
                if (r >= 1) fcand_consider(best, r, colv, sm.rhs[tid], precision);
            }
            __syncthreads();
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
        s.f_pc = cn;
        s.f_final_buf = in_buf ^ 1;
        if (cn == 0) {  // optimal after this pivot
            s.status = ST_DONE; s.optimal = 1; s.do_pivot = 0;
            s.obj_cell = n0.x;  // thread 0 owns column 0 of the updated cost row
        }
    }
}

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
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        copy_state(f.c.st, fin);
    }
}

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
#define JSLP_R_MAXROWS 16   // ... of the tall geometry (512 lanes x 4 columns x 16 rows: H <= 16 * 256)
#define JSLP_R_GRAN 8        // 8-byte granules per workgroup summary (7 used)
typedef unsigned long long u64_t;

struct ResCtx {
    Ctx c;
    u64_t* gran[2];       // [G][8] data-tagged granules {tag = epoch + 1 : 32 | payload : 32}: q, kq, kdeg halves, rows
    u64_t* rows_pub[2];   // [G][ld] candidate rows (doubles as 8-byte words)
    u64_t* rowflag[2];    // [G] epoch tag: the workgroup's candidate row of that epoch is fully written through
    unsigned* abort_flag; // set when any spin gives up
    u64_t* decision[2];   // leader's per-pivot decision: 3 tagged granules {pr | stop << 16}, {quot lo}, {quot hi}
    u64_t* verdict[2];    // phase 1 only: leader's cycle-check verdict {tag | stop} (the entering column is known late there)
    u64_t* gor[2];        // [G] rare slow path: tagged per-workgroup flags for a chip-wide OR
    int32_t G, rpb, H;
    int32_t iters_cap;
    u64_t* dbg;  // JSLP_DEBUG_RESIDENT builds only
};

#define AG_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define AG_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define JSLP_SPIN_LIMIT (1u << 22)
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
    int32_t pubrow;
    unsigned dec[4];
    double xq[2];   // phase 1: quot and k0 broadcast by the lane pair owning column pc
    double quo[JSLP_R_MAXROWS];
    int32_t kind[JSLP_R_MAXROWS];
    double col[JSLP_R_MAXROWS];  // my rows' entries in the pivot column / in column 0
    double rhs[JSLP_R_MAXROWS];
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
        if (spins > JSLP_SPIN_LIMIT) { if ((w & 63) == 0) AG_STORE(f.abort_flag, 1u); return false; }
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

// Pricing (simplex.ts:118-219, no unrestricted variables) of the cost-row pair (columns c0, c0+1) each lane
// holds, reduced with three LDS atomics: first batch holding a candidate, best value in it, first column with
// that value.  Positive doubles order like their bit patterns.  Returns the column (0 = none) and its value.
// `sm.p_*` must have been reset (p_batch = INT_MAX, p_val = 0, p_col = INT_MAX) before a preceding barrier.
template <int CPT>
__device__ __forceinline__ int price_row_lds(const double (&x)[CPT], int c0, const int (&pb)[CPT], const Ctx& c, RSmem& sm,
                                             double* value) {
    double bv = c.precision;
    int bi = 0, bb = 0;
#pragma unroll
    for (int j = 0; j < CPT; j++) {  // my columns in order: earlier batch first, bigger value inside a batch, first index on ties
        const int col = c0 + j;
        const bool ok = col >= 1 && col < c.W && x[j] > c.precision;
        const bool take = ok && (bi == 0 || pb[j] < bb || (pb[j] == bb && x[j] > bv));
        bv = take ? x[j] : bv;
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
    *value = __longlong_as_double((long long)wv);
    return sm.p_col;
}

#ifdef JSLP_DEBUG_RESIDENT
#define RT_MARK(i) do { const u64_t _now = __builtin_amdgcn_s_memtime(); rt_acc[i] += _now - rt_prev; rt_prev = _now; } while (0)
#else
#define RT_MARK(i) do { } while (0)
#endif

__device__ __forceinline__ void reset_reductions(RSmem& sm) {  // one thread, before a barrier
    sm.p_batch = 0x7fffffff; sm.p_val = 0; sm.p_col = 0x7fffffff;
    sm.l_rdeg = 0x7fffffff; sm.l_q = ~0ull; sm.l_r = 0x7fffffff;
}

// Chip-wide OR of one flag per workgroup (rare slow path of phase 1, see the lazily-zeroed pivot-row entries):
// every workgroup publishes a tagged granule and polls everybody else's.  Returns -1 on abort.
__device__ __forceinline__ int global_or(const ResCtx& f, int par, unsigned tag, int flag, RSmem& sm) {
    const int tid = threadIdx.x, b = blockIdx.x;
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
            if (spins > JSLP_SPIN_LIMIT) { AG_STORE(f.abort_flag, 1u); ok = 0; break; }
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
    double k0;
    int pc, end_code, unbounded_col, hist_n, it1, it2;
    unsigned epoch;
    long long trace_n;
#ifdef JSLP_DEBUG_RESIDENT
    u64_t rt_acc[8];
    u64_t rt_prev;
#endif
};

// One phase of the solve.  PHASE is a compile-time constant so that the phase-2 loop -- the hot one -- carries none of
// the phase-1 branches; returns when the solve ends (R.end_code != 0) or, for PHASE == 1, when phase 1 is over
// (end_code stays 0 and the caller starts phase 2).
template <int PHASE, int CPT, int ROWS>
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
    while (end_code == 0) {
        if ((it1 - it1_start) + (it2 - it2_start) >= f.iters_cap) { end_code = 4; break; }
        const int par = epoch & 1;
        const unsigned tag = epoch + 1;
        RT_MARK(7);
        // ---- A: my rows' summary: phase 2 = ratio test for column pc (simplex.ts:276-296); phase 1 = most negative RHS
        //         below -precision (simplex.ts:39-49) -------------------------------------------------------------------
        bool has_pc = phase == 2 && colok && pc >= c0 && pc < c0 + CPT;
        if (has_pc) {  // (conditional stores, not selects among register-array elements: those end up in scratch)
#pragma unroll
            for (int j = 0; j < CPT; j++)
                if (pc == c0 + j) {
#pragma unroll
                    for (int i = 0; i < ROWS; i++) sm.col[i] = a[i][j];
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
                const double colv = sm.col[tid], rhs = sm.rhs[tid];
                int kind = 0;  // 0 skip, 1 degenerate winner, 2 quotient candidate (phase 1: RHS candidate)
                double quo = 0.0;
                if (phase == 1) {
                    if (r >= 1 && r < r_end && rhs < -precision) { quo = rhs; kind = 2; }
                } else if (r >= 1 && r < r_end && !(-precision < colv && colv < precision)) {
                    if (colv > 0 && precision > rhs && rhs > -precision) kind = 1;
                    else { quo = rhs / colv; kind = quo > precision ? 2 : 0; }
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
                const double quo = sm.quo[i], colv = sm.col[i];
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
            u64_t* rp = f.rows_pub[par] + (long long)b * ld + c0;
#pragma unroll
            for (int i = 0; i < ROWS; i++)
                if (r_begin + i == pubrow) {  // uniform
#pragma unroll
                    for (int j = 0; j < CPT; j++) AG_STORE(rp + j, (u64_t)__double_as_longlong(a[i][j]));
                }
        }
        RT_MARK(1);
        // ---- C: workgroup 0 is the LEADER: its last four waves gather everybody's tagged summaries (data = flag)
        //         while all other waves, everywhere, drain their row stores -----------------------------------------------
        bool swept = true;
        SweptCand sc;
        sc.qbits = ~0ull; sc.kq = 0; sc.kdeg = 0; sc.r = 0; sc.rdeg = 0x7fffffff;
        const bool sweeper = b == 0 && tid >= sweep0;
        if (sweeper) swept = sweep_summary(f, par, tag, tid - sweep0, sc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int all_swept = __syncthreads_and(swept ? 1 : 0);
        if (!all_swept) { end_code = 5; break; }
        if (tid == 0 && pubrow != 0) AG_STORE(f.rowflag[par] + b, (u64_t)tag);  // every wave has drained: row is visible
        RT_MARK(2);
        // ---- D: the leader decides (winner, unboundedness, cycle check) and broadcasts three tagged granules -----------
        int pr = 0, stop = 0;
        double quot = 0.0;
        if (b == 0) {
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
                const unsigned payload = tid == 0 ? ((unsigned)pr | ((unsigned)stop << 16)) : (tid == 1 ? (unsigned)qb : (unsigned)(qb >> 32));
                AG_STORE(f.decision[par] + tid, ((u64_t)tag << 32) | payload);
            }
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
                    if (spins > JSLP_SPIN_LIMIT) { if (tid == 0) AG_STORE(f.abort_flag, 1u); ok = 0; break; }
                }
                if (tid < 3) sm.dec[tid] = (unsigned)x;
                if (tid == 0) sm.ok = ok;
            }
            __syncthreads();
            if (!sm.ok) { end_code = 5; break; }
            pr = (int)(sm.dec[0] & 0xffffu);
            stop = (int)(sm.dec[0] >> 16);
            quot = __longlong_as_double((long long)((u64_t)sm.dec[1] | ((u64_t)sm.dec[2] << 32)));
        }
        RT_MARK(3);
        if (stop == 3) { end_code = 2; unbounded_col = pc; break; }
        if (stop == 1) { end_code = 3; break; }
        if (stop == 2) { end_code = 6; break; }
        if (stop == 4) {  // phase 1 is over: the caller starts phase 2 with a fresh history (simplex.ts:14-23, 102)
            hist_n = 0;
            epoch += 1;
            return;
        }
        // ---- E: the winning row: loaded speculatively together with its flag; re-loaded in the rare case the flag
        //         (which follows the winner's drain) was not up yet ------------------------------------------------------
        const int bw = pr / f.rpb;
        const u64_t* rp_in = f.rows_pub[par] + (long long)bw * ld + c0;
        double pv[CPT];
#pragma unroll
        for (int j = 0; j < CPT; j++) pv[j] = 0.0;
        for (;;) {
            u64_t flag = 0;
            if (tid == 0) flag = AG_LOAD(f.rowflag[par] + bw);
            if (colok) {
#pragma unroll
                for (int j = 0; j < CPT; j++) pv[j] = __longlong_as_double((long long)AG_LOAD(rp_in + j));
            }
            if (tid == 0) {
                int ok = 1;
                if ((unsigned)flag != tag) {
                    unsigned spins = 0;
                    ok = 2;  // row must be re-read once the flag is up
                    while ((unsigned)AG_LOAD(f.rowflag[par] + bw) != tag) {
                        __builtin_amdgcn_s_sleep(1);
                        ++spins;
                        if ((spins & 63u) == 0 && AG_LOAD(f.abort_flag) != 0u) { ok = 0; break; }
                        if (spins > JSLP_SPIN_LIMIT) { AG_STORE(f.abort_flag, 1u); ok = 0; break; }
                    }
                }
                sm.ok = ok;
            }
            __syncthreads();
            const int okv = sm.ok;
            __syncthreads();
            if (okv == 2) continue;
            if (okv == 0) end_code = 5;
            break;
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
                if (col >= 1 && col < W && coef < -precision) {
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
            if (c.check_cycles) {  // simplex.ts:78-93: only now is the (leaving, entering) pair known
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
                            if (spins > JSLP_SPIN_LIMIT) { AG_STORE(f.abort_flag, 1u); break; }
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
            const int g = global_or(f, par, tag, local_any, sm);
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
                const double nv = -k0 / quot;
#pragma unroll
                for (int j = 0; j < CPT; j++)
                    if (pc == c0 + j) r0[j] = nv;
            }
        }
#pragma unroll
        for (int i = 0; i < ROWS; i++) {
            const int r = r_begin + i;
            if (r >= r_end) continue;
            if (r == 0) {  // workgroup 0 owns the cost row
#pragma unroll
                for (int j = 0; j < CPT; j++) a[i][j] = r0[j];
                continue;
            }
            if (r == pr) {
#pragma unroll
                for (int j = 0; j < CPT; j++) a[i][j] = p[j];
                continue;
            }
            const double ki = sm.col[i];  // pivot-column entry of row i (still in LDS from step A)
            if (nonzero16(ki)) {
#pragma unroll
                for (int j = 0; j < CPT; j++)
                    if (nz[j]) a[i][j] = eliminate(a[i][j], ki, p[j]);
                if (has_pc) {
                    const double nv = -ki / quot;
#pragma unroll
                    for (int j = 0; j < CPT; j++)
                        if (pc == c0 + j) a[i][j] = nv;
                }
            }
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
        trace_n += 1;
        if (phase == 1) it1 += 1; else it2 += 1;
        epoch += 1;
        RT_MARK(6);
        // ---- G: phase 2: price the new cost row -> entering column of the next pivot -----------------------------------
        if (phase == 2) {
            pc = price_row_lds<CPT>(r0, c0, pb, c, sm, &k0);
            if (pc == 0) end_code = 1;
        }
    }
}

// THREADS x CPT >= ld: <1024, 2> = lane pairs of columns, 4 waves per SIMD; <512, 4> = half the waves to synchronise,
// twice the independent work per lane (and 256 VGPRs per lane).
template <int THREADS, int CPT, int ROWS>
__global__ void __launch_bounds__(THREADS) k_simplex_resident(ResCtx f) {
    static_assert(ROWS <= JSLP_R_MAXROWS, "RSmem holds one entry per row of the workgroup");
    __shared__ RSmem sm;
    ResRegs<CPT, ROWS> R;
#ifdef JSLP_DEBUG_RESIDENT
    for (int i = 0; i < 8; i++) R.rt_acc[i] = 0;
    R.rt_prev = __builtin_amdgcn_s_memtime();
#endif
    const Ctx& c = f.c;
    const int tid = threadIdx.x, b = blockIdx.x;
    const int ld = c.ld, W = c.W, H = f.H;
    const double precision = c.precision;
    const int c0 = tid * CPT;
    const bool colok = c0 < ld;
    const int r_begin = b * f.rpb, r_end = min(H, r_begin + f.rpb);
    DevState* st = c.st;
    static_assert(CPT % 2 == 0, "lanes load and store their columns as 16-byte pairs");

    // ---- load my rows and the cost row into registers ---------------------------------------------------
    double (&a)[ROWS][CPT] = R.a;
    double (&r0)[CPT] = R.r0;
#pragma unroll
    for (int j = 0; j < CPT; j += 2) {
        double2 t = make_double2(0, 0);
        if (colok) t = *reinterpret_cast<const double2*>(c.A + c0 + j);
        r0[j] = t.x; r0[j + 1] = t.y;
    }
#pragma unroll
    for (int i = 0; i < ROWS; i++) {
        const int r = r_begin + i;
        const bool mine = i < f.rpb && r < r_end && colok;
#pragma unroll
        for (int j = 0; j < CPT; j += 2) {
            double2 t = make_double2(0, 0);
            if (mine) t = *reinterpret_cast<const double2*>(c.A + (long long)r * ld + c0 + j);
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
    if (tid == 0) reset_reductions(sm);
    __syncthreads();
    // pricing batch of my columns (simplex.ts:118-127): fixed for the whole solve
    int pb[CPT];
#pragma unroll
    for (int j = 0; j < CPT; j++) pb[j] = c.use_partial && c0 + j >= 1 ? (c0 + j - 1) / c.batch : 0;
    R.k0 = 0.0;  // reduced cost of the entering column = cost-row entry of column pc
    R.pc = 0;
    R.end_code = 0;  // 1 optimal, 2 unbounded, 3 cycle, 4 iteration cap, 5 aborted hand-off, 6 history full, 7 infeasible
    R.unbounded_col = 0;
    R.epoch = 0;
    if (phase == 1) {
        resident_phase<1, CPT, ROWS>(f, sm, R, it1_start, it2_start, pb);
        if (R.end_code == 0) phase = 2;
    }
    if (R.end_code == 0) {  // phase 2 (simplex.ts:100-325): first entering column, then the hot loop
        R.pc = price_row_lds<CPT>(r0, c0, pb, c, sm, &R.k0);
        if (R.pc == 0) R.end_code = 1;
        else resident_phase<2, CPT, ROWS>(f, sm, R, it1_start, it2_start, pb);
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
                    *reinterpret_cast<double2*>(c.A + (long long)r * ld + c0 + j) = make_double2(a[i][j], a[i][j + 1]);
            }
        }
    }
    if (b == 0 && tid == 0) {
        st->it1 = it1;
        st->it2 = it2;
        st->trace_n = trace_n;
        st->hist_n = hist_n;
        st->iters_left -= (it1 - it1_start) + (it2 - it2_start);
        st->do_pivot = 0;
        st->status = ST_DONE;
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
