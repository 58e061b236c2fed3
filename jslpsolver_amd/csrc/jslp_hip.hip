// jslp_hip.hip -- host side of the MI355X engine: the C ABI of include/jslp_engine.h over the kernels in
// jslp_kernels.hip.h.  Build (see __graft_entry__.build):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -o libjslp_hip.so jslp_hip.hip
//
// Launch strategy (DESIGN.md "Kernels"):
//   * large tableau  -> per pivot one k_select (1 workgroup) + one k_update (whole chip) on the engine's
//     stream; pivots are enqueued in chunks and the 128-byte device state is polled once per chunk, so the
//     GPU never waits for the host inside a chunk.  Kernels launched after the solve ended exit at once.
//   * small tableau / batch of branch-and-bound nodes -> k_simplex_wg: one workgroup runs the whole
//     simplex() of one tableau copy ("slot"); a batch is one launch with grid = #nodes.
// (see xl_fits below: the XCD-local kernels live in the test library; defined up here because jslp_engine_create reads it first)
#if (defined(JSLP_CHAOS_BUILD) || defined(JSLP_DEV_XL_ONLY)) && !defined(JSLP_WITH_XL)
#define JSLP_WITH_XL 1
#endif
#include "../../include/jslp_engine.h"
#include "jslp_kernels.hip.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

static thread_local char g_err[512];
static int fail(int code, const char* fmt, const char* a = "", const char* b = "") {
    snprintf(g_err, sizeof g_err, fmt, a, b);
    return code;
}
#define HIPC(expr)                                                                              \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) return fail(JSLP_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

static const size_t WGLDS_MAX_BYTES = 64 * 1024;  // largest dynamic LDS the LDS-resident one-workgroup kernels are launched with
struct jslp_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;  // read-back of a finished group of nodes while the next group computes
    hipEvent_t ev_group = nullptr;
    int32_t H0 = 0, W = 0, ld = 0, cap_rows = 0, n_idx = 0;
    double precision = 1e-8;
    int32_t batch = 50, use_partial = 0;
    int uploaded = 0, has_save = 0;
    int slot0_synced = 0;  // slot 0 = current snapshot except for its dirty rows (st.gen == st.s_gen): k_node_wg may be used
    int slots_synced = 0;  // slots 1 .. slots_synced-1 are in that state too (only slot 0 ever leaves it on its own)
    double evaluation = 0;
    // slots (slot 0 = the live tableau).  All per-slot arrays live in ONE device allocation (slot_arena), the
    // snapshot / flags / trace in another (static_arena): hipMalloc / hipFree cost ~0.1 ms apiece, and a Solve of a
    // small model creates and destroys an engine
    int n_slots = 0;
    Slots s{};
    char* slot_arena = nullptr; size_t slot_bytes = 0;
    char* static_arena = nullptr; size_t static_bytes = 0;
    char* spare_slot_arena = nullptr; size_t spare_slot_bytes = 0;  // handed over by the resource pool at create()
    // snapshot
    double* snap_A = nullptr; double* snap_rhs = nullptr;
    double* snap_AT = nullptr; int snap_ldT = 0;  // the saved root transposed (tableaus the LDS node kernels take; else nullptr)
    int32_t *snap_vibr = nullptr, *snap_vibc = nullptr, *snap_rbv = nullptr, *snap_cbv = nullptr;
    uint8_t* d_unr = nullptr;
    uint8_t* d_isint = nullptr;  // variable.isInteger per variable index (MIR cuts)
    int32_t n_opt = 0; double* snap_oo = nullptr;  // optional objectives (slot copies live in s.oo)
    // cuts staging
    // cuts staging: ONE pinned host buffer -> ONE device buffer per call: [value | offs | var | type]
    char* d_cuts = nullptr; char* h_cuts = nullptr; size_t cuts_bytes = 0;
    int32_t* d_cut_offs = nullptr; int8_t* d_cut_type = nullptr; int32_t* d_cut_var = nullptr; double* d_cut_val = nullptr;
    int32_t* d_cut_order = nullptr; int* d_queue = nullptr;  // batch hand-out order (most cuts first) + queue counter, uploaded with the cuts
    std::vector<int32_t> order_scratch;
    bool queue_wgs_opt = false; int node_queue_launches = 0; long long resident_fetch_retries = 0;  // (which build the count below is for; batches that went through k_node_queue)
    int queue_wgs = 0; size_t queue_wgs_lds = 0;             // resident workgroups of k_node_queue<512> for this LDS size
    // read-back staging (device + pinned host)
    // read-back staging: ONE device buffer -> ONE pinned buffer per group: [states | rhs | rows]
    char* d_out = nullptr; char* h_out = nullptr;
    double* d_rhs = nullptr; int32_t* d_rows = nullptr; DevState* d_states = nullptr;
    double* h_rhs = nullptr; int32_t* h_rows = nullptr; DevState* h_states = nullptr;
    size_t out_bytes_cap = 0;
    DevState* h_state = nullptr;  // pinned, 1 entry, followed by the completion flag of the one-launch node kernel
    unsigned done_seq = 0;
    // upload staging (pinned): [matrix, row stride W | vibr | vibc | unrestricted list]; the matrix part is what
    // jslp_engine_host_matrix hands to the host to build the tableau in; d_up = device twin of the blob
    // (and of the matrix when it needs the W -> ld repack)
    char* h_up = nullptr; size_t h_up_bytes = 0;
    char* d_up = nullptr; size_t d_up_bytes = 0;
    int host_matrix_out = 0;  // the host holds a pointer into h_up
    // the one-call-only redirection of the read-back (device pool: every member copies straight into the pool's buffer)
    DevState* ext_states = nullptr; double* ext_rhs = nullptr; int32_t* ext_rows = nullptr;
    // jslp_engine_relax_batch_device: the caller's DEVICE buffers take the outcomes (no host copy at all)
    DevState* dev_states = nullptr; double* dev_rhs = nullptr; int32_t* dev_rows = nullptr; int32_t dev_stride = 0;
    // compact read-back (jslp_engine_relax_watched)
    int32_t* d_watch = nullptr; int32_t n_watch = 0;
    // work counters
    int counting = 0;
    cnt_t* d_cnt = nullptr;
    jslp_work_counters wc{};
    // snapshot generation as the device pool sees it: bumped by save() and upload()
    unsigned long long root_seq = 0;
    // safety net of the register-resident kernel: copy of slot 0 taken before the cooperative launch
    char* r_arena = nullptr; size_t r_arena_bytes = 0;  // hand-off buffers + this backup in ONE allocation (parked in the resource pool)
    DevState* r_backup_st = nullptr;
    double* rb_A = nullptr; int32_t *rb_vibr = nullptr, *rb_vibc = nullptr, *rb_rbv = nullptr, *rb_cbv = nullptr;
    int* d_done_count = nullptr;  // workgroups of a small batch that have delivered their outcome (k_node_lds: the last one raises the completion flag)
    unsigned long long* d_nnz = nullptr; long long nnz = -1;  // non-zero cells of the uploaded tableau (counted on the device)
    unsigned spin_limit = 0; int test_abort_epoch = -1; int test_late_wave0 = 0;
    int resident_fallbacks = 0;  // solves that were rolled back and re-run through the streaming kernels
    int resident_handovers = 0;  // solves the lean resident kernel handed to the general one (cycle-check history beyond its LDS copy)
    int resident_launches = 0;   // cooperative launches of k_simplex_resident the runtime accepted
    int resident_refusals = 0;   // ... it refused (or that no build exists for): the solve went through the streaming kernels
    double dev_prev_evaluation = 0.0;  // jslp_engine_relax_batch_device: the evaluation its nodes started from (results_from_states)
    int dev_prev_valid = 0;
    // checkpoints (incremental-branch-and-cut.ts:31-44): equally sized device buffers, recycled through a free list
    struct Ckpt {
        char* mem = nullptr;
        double* A = nullptr; double* rhs = nullptr;
        int32_t *vibr = nullptr, *vibc = nullptr, *rbv = nullptr, *cbv = nullptr;
        int32_t H = 0, last_element_index = 0;
        double evaluation = 0;
        int live = 0;
    };
    std::vector<Ckpt> ckpts;
    std::vector<char*> ck_free;
    // fp32 twin of slot 0 (jslp_engine_simplex_f32), allocated on first use
    f32::Slots s32{};
    char* arena32 = nullptr;
    // policy
    int force_path = 0;  // 0 auto, 1 workgroup kernel, 2 select+update kernels only, 3 fused phase 2
    int force_xl = 0, xl_on = 0;  // JSLP_FORCE_PATH=xl / JSLP_XL=0: the XCD-local register-resident geometry (resident_geometry 6)
    int force_resident = 0;  // JSLP_FORCE_PATH=resident: also the register-resident geometries the default policy leaves to the streaming kernels
    int32_t n_unr = 0;
    int nt = 0;  // JSLP_NT=1: non-temporal hints in the fused kernel
    // fused phase-2 pipeline (ping-pong buffer + per-workgroup candidates)
    double* f_buf1 = nullptr; FCand* f_cands[2] = {nullptr, nullptr}; double* f_pcol[2] = {nullptr, nullptr};
    uint8_t* f_uflags = nullptr;  // fused pipeline with unrestricted variables: 2 x (column flags | row flags)
    double* f_oo1 = nullptr; int f_oo1_rows = 0;  // fused pipeline: the optional objectives' second (ping-pong) buffer
    long long p1_slow_pivots = 0;  // pivots the fused phase 1 handed to k_select + k_update (tiny pivot-row entries, see k_fused_p1)
    DevState* f_st[2] = {nullptr, nullptr};
    // register-resident phase 2 (one cooperative launch): hand-off buffers
    u64_t* r_gran = nullptr;  // [2][G][8] granules then [2][G] row flags (one allocation, zeroed per launch)
    int32_t max_uploaded_idx = -1;  // largest variable index in the row / column maps of the last upload()
    int abort_injected = 0;  // JSLP_INJECT_RESIDENT_ABORT_US fired in this simplex() already
    unsigned* h_abort_ptr = nullptr;  // pinned word the lean resident kernels look at every 1024 pivots (host-requested abort)
    size_t r_slot_doubles = 0;  // doubles per candidate-row slot the arena was carved for
    u64_t* r_rows[2] = {nullptr, nullptr}; unsigned* r_sync = nullptr; int2* r_hist_all = nullptr; int r_want_hist = 0; Ctx* r_ctx_dev = nullptr;
    int no_resident = 0;
    int res_cpt = 2;  // columns per lane of the resident kernel (JSLP_RES_CPT=2|4)
    int one_launch_nodes = 1;  // JSLP_NO_NODE_KERNEL=1: single children go through the five-launch sequence
    int use_wglds = 1;         // JSLP_NO_WGLDS=1: the generic one-workgroup kernels (global-memory selection state) only
    const char* last_path = "none";
    // timing
    int timing = 0;
    double upd_ms = 0, total_ms = 0;
    long long upd_launches = 0;
    std::vector<hipEvent_t> ev_pool;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
};

static const long long WG_CELLS_SINGLE = 64 * 1024;         // one workgroup beats 2 launches/pivot below this
static const long long WG_CELLS_BATCH = 4LL * 1024 * 1024;  // batches use one workgroup per node up to this
#ifndef JSLP_SMALL_BATCH_1024_DEFAULT
#define JSLP_SMALL_BATCH_1024_DEFAULT 256  // batches of at most this many nodes (one per CU) take the 1024-thread node kernel: 16 nodes 111 -> 102 us, 64 nodes 147 -> 132 us (r03_o); 0 = never
#endif
static int wg_batch_threads() {  // workgroup size of the per-node kernel when a call carries several nodes
    static int v = -1;
    if (v < 0) {
        const char* t = getenv("JSLP_WG_BATCH_THREADS");  // tuning knob (tools/relax_batch_scaling.py)
        v = t ? atoi(t) : 512;  // measured on Monster_II node batches (tools/wg_shape_sweep.sh): 512 > 1024 > 256
        if (v != 256 && v != 1024) v = 512;
    }
    return v;
}
static int group_max() {  // nodes per group of a batch (= tableau copies alive at once)
    static int v = -1;
    if (v < 0) {
        const char* t = getenv("JSLP_GROUP_MAX");  // tuning knob
        v = t ? atoi(t) : 1024;
        if (v < 1) v = 1024;
    }
    return v;
}
static int zero_copy() {  // batch outcomes written by the kernels straight into the pinned read-back buffer (no copy stream)
    static int v = -1;
    if (v < 0) {
        const char* t = getenv("JSLP_ZERO_COPY");  // tuning knob
        v = t ? atoi(t) : 1;
    }
    return v;
}
static int snapshot_transpose_on() {  // pivot-column reads of untouched rows from the transposed root
    static int v = -1;
    if (v < 0) {
        const char* t = getenv("JSLP_SNAPSHOT_TRANSPOSE");  // tuning knob
        v = t ? atoi(t) : 1;
    }
    return v;
}
static int batch_poll_on() {  // a one-group batch of the 1024-thread node kernel ends by polling its completion flag (JSLP_BATCH_POLL=0: stream synchronisation)
    static int v = -1;
    if (v < 0) { const char* t = getenv("JSLP_BATCH_POLL"); v = (t && t[0] == '0') ? 0 : 1; }
    return v;
}
static int node_cow_single() {  // single-node call: copy-on-write start (JSLP_NODE_COW_SINGLE=0: restore eagerly, as the batch groups do)
    static const int v = getenv("JSLP_NODE_COW_SINGLE") ? atoi(getenv("JSLP_NODE_COW_SINGLE")) : 1;
    return v;
}
static int node_cow() {  // queue kernel: copy-on-write slots (no restore between nodes)
    static int v = -1;
    if (v < 0) {
        const char* t = getenv("JSLP_NODE_COW");  // tuning knob
        v = t ? atoi(t) : 1;
    }
    return v;
}
static int node_queue() {  // a batch in one launch of resident workgroups pulling nodes from a queue
    static int v = -1;
    if (v < 0) {
        // tuning knob: 0 = one launch per group, 1 = the queue hands the nodes out in call order, 2 (default since round 6) = most cuts first.
        // The cut count predicts a node's repair pivots, and 2416 nodes on 768 resident workgroups are 3.15 rounds: longest-first keeps the last
        // workgroups to finish on the cheap nodes.  Round 2 measured it SLOWER (2.73 M/s against 2.94 M in call order: a node then cost ~3x what it
        // costs now and the order array's extra trip per node showed); re-measured on round 6's kernels, same session, compact read-back,
        // tools/wglds_timing.py: 4.94-5.06 M relaxations/s against 4.14-4.41 M in call order (+14 %); outcome digests identical (tools/queue_check.py)
        const char* t = getenv("JSLP_NODE_QUEUE");
        v = t ? atoi(t) : 2;
    }
    return v;
}
static long long wg_cells_child() {  // a single B&B child (few repair pivots) stays in one workgroup up to this
    static long long v = -1;
    if (v < 0) {
        const char* t = getenv("JSLP_WG_CELLS_CHILD");  // tuning knob (tools/child_path_times.py)
        v = t ? atoll(t) : 1536LL * 1024;
    }
    return v;
}
static const size_t HIST_CAP_MAIN = 1u << 20;
static const size_t HIST_CAP_SLOT = 1u << 16;
static const long long TRACE_CAP = 1LL << 20;

extern "C" const char* jslp_backend_name(void) { return "hip-gfx950"; }
extern "C" const char* jslp_last_error(void) { return g_err; }
extern "C" int jslp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int32_t round_up(int32_t x, int32_t m) { return (x + m - 1) / m * m; }
static dim3 copy_grid(const jslp_engine* e, int slots);

struct Carver {  // hands out 256-byte aligned pieces of one allocation; first pass (base == nullptr) just sizes it
    char* base;
    size_t off;
    template <class T>
    T* take(size_t count) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += sizeof(T) * count;
        return p;
    }
};

// ---- resource pool ---------------------------------------------------------------------------------------------------
// A Solve() of a small model creates and destroys an engine; stream / event / pinned-memory creation and the two device
// arenas cost ~4 ms per engine on this stack -- more than every pivot of the reference's fixtures.  Destroyed engines
// therefore park those resources here (a handful of entries, arenas up to 1 GiB each) and the next create() on the same
// device takes them over, re-carving the arenas when they are large enough.
struct PooledRes {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    DevState* h_state = nullptr;
    char* static_arena = nullptr; size_t static_bytes = 0;
    char* slot_arena = nullptr; size_t slot_bytes = 0;
    char* d_cuts = nullptr; char* h_cuts = nullptr; size_t cuts_bytes = 0;  // staging of the cut lists
    char* d_out = nullptr; char* h_out = nullptr; size_t out_bytes = 0;     // read-back staging
    char* h_up = nullptr; size_t h_up_bytes = 0; char* d_up = nullptr; size_t d_up_bytes = 0;  // upload staging
    char* r_arena = nullptr; size_t r_arena_bytes = 0;  // the resident kernel's hand-off buffers + backup
    hipStream_t copy_stream = nullptr; hipEvent_t ev_group = nullptr;  // read-back overlap of node batches
};
static std::mutex g_pool_mu;
static std::vector<PooledRes> g_pool;
static const size_t POOL_MAX_ENTRIES = 4;
static const size_t POOL_MAX_ARENA = (size_t)1 << 30;

static bool pool_enabled() {
    static int v = -1;
    if (v < 0) { const char* t = getenv("JSLP_NO_POOL"); v = (t && t[0] == '1') ? 0 : 1; }
    return v == 1;
}
static bool pool_take(int device, PooledRes* out) {
    if (!pool_enabled()) return false;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_pool.size(); i++)
        if (g_pool[i].device == device) {
            *out = g_pool[i];
            g_pool.erase(g_pool.begin() + (long)i);
            return true;
        }
    return false;
}
static bool pool_give(const PooledRes& r) {
    if (!pool_enabled() || r.static_bytes > POOL_MAX_ARENA || r.slot_bytes > POOL_MAX_ARENA || r.out_bytes > POOL_MAX_ARENA / 4 ||
        r.h_up_bytes > POOL_MAX_ARENA / 4 || r.r_arena_bytes > POOL_MAX_ARENA)
        return false;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (g_pool.size() >= POOL_MAX_ENTRIES) return false;
    g_pool.push_back(r);
    return true;
}

extern "C" void jslp_release_pooled_resources(void) {
    std::vector<PooledRes> drop;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        drop.swap(g_pool);
    }
    for (auto& r : drop) {
        hipSetDevice(r.device);
        hipFree(r.static_arena); hipFree(r.slot_arena); hipFree(r.d_cuts); hipFree(r.d_out);
        if (r.h_cuts) hipHostFree(r.h_cuts);
        if (r.h_out) hipHostFree(r.h_out);
        if (r.h_up) hipHostFree(r.h_up);
        hipFree(r.d_up); hipFree(r.r_arena);
        if (r.ev_group) hipEventDestroy(r.ev_group);
        if (r.copy_stream) hipStreamDestroy(r.copy_stream);
        if (r.h_state) hipHostFree(r.h_state);
        if (r.ev_begin) hipEventDestroy(r.ev_begin);
        if (r.ev_end) hipEventDestroy(r.ev_end);
        if (r.stream) hipStreamDestroy(r.stream);
    }
}

static void free_slots(jslp_engine* e, bool keep_arena = false) {
    if (!keep_arena) { hipFree(e->slot_arena); e->slot_bytes = 0; }
    hipFree(e->s.oo);
    e->slot_arena = nullptr;
    e->s.dirty = nullptr; e->s.oo = nullptr;
    e->s.A = nullptr; e->s.vibr = e->s.vibc = e->s.rbv = e->s.cbv = nullptr;
    e->s.prow = e->s.pcol = nullptr; e->s.rhs = nullptr; e->s.st = nullptr; e->s.hist = nullptr;
}

static void carve_slots(Slots& s, Carver& cv, int n) {
    s.A = cv.take<double>((size_t)s.A_stride * n);
    s.vibr = cv.take<int32_t>((size_t)s.vibr_stride * n);
    s.vibc = cv.take<int32_t>((size_t)s.vibc_stride * n);
    s.rbv = cv.take<int32_t>((size_t)s.idx_stride * n);
    s.cbv = cv.take<int32_t>((size_t)s.idx_stride * n);
    s.prow = cv.take<double>((size_t)s.prow_stride * n);
    s.pcol = cv.take<double>((size_t)s.pcol_stride * n);
    s.dirty = cv.take<uint8_t>((size_t)s.pcol_stride * n);
    s.rhs = cv.take<double>((size_t)s.pcol_stride * n);
    s.st = cv.take<DevState>((size_t)n);
    s.hist = cv.take<int2>((size_t)s.hist_cap * n);
}

// (re)allocate the slot arrays for n slots, preserving slot 0
static int ensure_slots(jslp_engine* e, int n) {
    if (n <= e->n_slots) return JSLP_OK;
    Slots o = e->s, s = e->s;
    char* old_arena = e->slot_arena;
    const int old_n = e->n_slots;
    s.A_stride = (long long)e->cap_rows * e->ld;
    s.vibr_stride = e->cap_rows;
    s.vibc_stride = e->W;
    s.idx_stride = e->n_idx;
    s.prow_stride = e->ld;
    s.pcol_stride = e->cap_rows;
    s.hist_cap = n == 1 ? (int32_t)HIST_CAP_MAIN : (int32_t)HIST_CAP_SLOT;
    s.ld = e->ld; s.W = e->W; s.batch = e->batch; s.use_partial = e->use_partial; s.precision = e->precision;
    s.has_unr = e->n_unr > 0 ? 1 : 0;
    s.unr = e->d_unr;
    Carver sizing{nullptr, 0};
    carve_slots(s, sizing, n);
    char* arena = nullptr;
    size_t arena_bytes = sizing.off + 256;
    if (e->spare_slot_arena && e->spare_slot_bytes >= arena_bytes) {  // from the resource pool
        arena = e->spare_slot_arena;
        arena_bytes = e->spare_slot_bytes;
        e->spare_slot_arena = nullptr;
    } else {
        HIPC(hipMalloc(&arena, arena_bytes));
    }
    Carver cv{arena, 0};
    carve_slots(s, cv, n);
    // slot 0's matrix is zero-filled by upload(); other slots are filled by their first (full) restore
    HIPC(hipMemsetAsync(s.dirty, 0, (size_t)s.pcol_stride * n, e->stream));
    HIPC(hipMemsetAsync(s.st, 0, sizeof(DevState) * n, e->stream));  // gen = 0: slots hold no snapshot copy yet
    s.n_opt = e->n_opt;
    s.oo_stride = (long long)e->n_opt * e->ld;
    s.oo = nullptr;
    if (e->n_opt > 0) {
        HIPC(hipMalloc(&s.oo, sizeof(double) * (size_t)s.oo_stride * n));
        HIPC(hipMemsetAsync(s.oo, 0, sizeof(double) * (size_t)s.oo_stride * n, e->stream));
    }
    if (old_n > 0) {  // carry the live tableau over
        HIPC(hipMemcpyAsync(s.A, o.A, sizeof(double) * o.A_stride, hipMemcpyDeviceToDevice, e->stream));
        HIPC(hipMemcpyAsync(s.vibr, o.vibr, sizeof(int32_t) * o.vibr_stride, hipMemcpyDeviceToDevice, e->stream));
        HIPC(hipMemcpyAsync(s.vibc, o.vibc, sizeof(int32_t) * o.vibc_stride, hipMemcpyDeviceToDevice, e->stream));
        HIPC(hipMemcpyAsync(s.rbv, o.rbv, sizeof(int32_t) * o.idx_stride, hipMemcpyDeviceToDevice, e->stream));
        HIPC(hipMemcpyAsync(s.cbv, o.cbv, sizeof(int32_t) * o.idx_stride, hipMemcpyDeviceToDevice, e->stream));
        HIPC(hipMemcpyAsync(s.st, o.st, sizeof(DevState), hipMemcpyDeviceToDevice, e->stream));
        HIPC(hipMemcpyAsync(s.dirty, o.dirty, (size_t)o.pcol_stride, hipMemcpyDeviceToDevice, e->stream));
        HIPC(hipMemcpyAsync(s.rhs, o.rhs, sizeof(double) * (size_t)o.pcol_stride, hipMemcpyDeviceToDevice, e->stream));
        if (e->n_opt > 0 && o.oo) HIPC(hipMemcpyAsync(s.oo, o.oo, sizeof(double) * (size_t)o.oo_stride, hipMemcpyDeviceToDevice, e->stream));
        HIPC(hipStreamSynchronize(e->stream));
        hipFree(old_arena);
        hipFree(o.oo);
    }
    e->s = s;
    e->slot_arena = arena;
    e->slot_bytes = arena_bytes;
    e->n_slots = n;
    e->slots_synced = std::min(e->slots_synced, 1);  // only slot 0 was carried over
    if (e->spare_slot_arena) { hipFree(e->spare_slot_arena); e->spare_slot_arena = nullptr; }  // too small: not needed any more
    return JSLP_OK;
}

// forget every checkpoint; their buffers go to the free list (or back to the driver)
static void drop_checkpoints(jslp_engine* e, int free_memory) {
    for (auto& c : e->ckpts)
        if (c.live) e->ck_free.push_back(c.mem);
    e->ckpts.clear();
    if (free_memory) {
        for (char* m : e->ck_free) hipFree(m);
        e->ck_free.clear();
    }
}

extern "C" int jslp_engine_create(jslp_engine** out, int device, int32_t height, int32_t width, int32_t row_capacity,
                                  double precision) {
    if (!out || height < 1 || width < 1 || row_capacity < height) return fail(JSLP_ERR_ARG, "create: bad dimensions");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(JSLP_ERR_DEVICE, "create: no HIP device visible (this engine has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(JSLP_ERR_ARG, "create: device ordinal out of range");
    HIPC(hipSetDevice(device));
    jslp_engine* e = new jslp_engine();
    e->device = device;
    e->H0 = height; e->W = width; e->cap_rows = row_capacity; e->precision = precision;
    e->ld = round_up(width, 16);
    e->n_idx = width + 2 * row_capacity + 2;
    // partial pricing parameters (simplex.ts:118-127)
    const int32_t n_columns = width - 1;
    int32_t b = (int32_t)floor(sqrt((double)n_columns));
    b = std::min(500, std::max(50, b));
    e->batch = b;
    e->use_partial = n_columns > b * 2;
    const char* fp = getenv("JSLP_FORCE_PATH");
    if (fp && !strcmp(fp, "wg")) e->force_path = 1;
    if (fp && !strcmp(fp, "sp")) e->force_path = 2;
    if (fp && !strcmp(fp, "fused")) { e->force_path = 3; e->no_resident = 1; }
    if (fp && !strcmp(fp, "resident")) { e->force_path = 3; e->force_resident = 1; }
#ifdef JSLP_WITH_XL
    if (fp && !strcmp(fp, "xl")) { e->force_path = 3; e->force_resident = 1; e->force_xl = 1; }  // the XCD-local resident geometry whatever the size (<= 1024 x 1024)
#else
    if (fp && !strcmp(fp, "xl")) { delete e; return fail(JSLP_ERR_UNSUPPORTED, "JSLP_FORCE_PATH=xl: the XCD-local kernels are compiled into the test library only (libjslp_hip_chaos.so, or build with -DJSLP_WITH_XL)"); }
#endif
    // JSLP_XL=1: the XCD-local geometry for every tableau it takes.  OFF by default: as measured in round 4 (profiles/r04_xl_*) it is
    // correct on every golden but not faster than what these sizes had -- 6.4-6.8 us per pivot on dense 501 x 501 / 1001 x 1001 against
    // 6.0 chip-wide, 8.6 against 6.9 (one LDS workgroup) on the sparse Monster LP: its pivot is bound by the instruction stream of
    // the 32-row unrolled loops, not by the hand-offs it shortens
    const char* xl = getenv("JSLP_XL");
    e->xl_on = (xl && xl[0] == '1') ? 1 : 0;
    const char* nr = getenv("JSLP_NO_RESIDENT");
    if (nr && nr[0] == '1') e->no_resident = 1;
    const char* rcpt = getenv("JSLP_RES_CPT");
    if (rcpt && rcpt[0] == '4') e->res_cpt = 4;
    if (rcpt && rcpt[0] == '2') e->res_cpt = 2;
    const char* nk = getenv("JSLP_NO_NODE_KERNEL");
    if (nk && nk[0] == '1') e->one_launch_nodes = 0;
    const char* nl = getenv("JSLP_NO_WGLDS");
    if (nl && nl[0] == '1') e->use_wglds = 0;
    const char* nt = getenv("JSLP_NT");
    e->nt = (nt && nt[0] == '1') ? 1 : 0;
    const char* sl = getenv("JSLP_SPIN_LIMIT");  // polls before a hand-off of the resident kernel gives up (tests shorten it)
    e->spin_limit = sl ? (unsigned)std::max(1LL, atoll(sl)) : JSLP_SPIN_LIMIT_DEFAULT;
    const char* ta = getenv("JSLP_TEST_RESIDENT_ABORT");  // tests only: abort the resident kernel's hand-off at this pivot
    e->test_abort_epoch = ta ? atoi(ta) : -1;
    const char* tl = getenv("JSLP_TEST_RESIDENT_LATE_WAVE0");  // tests only: wave 0 of every workgroup reaches each row fetch late
    e->test_late_wave0 = tl ? atoi(tl) : 0;
    int rc = JSLP_OK;
    PooledRes pooled;
    const bool have = pool_take(device, &pooled);
    auto init = [&]() -> int {
        if (have) {
            e->stream = pooled.stream; e->ev_begin = pooled.ev_begin; e->ev_end = pooled.ev_end; e->h_state = pooled.h_state;
            e->spare_slot_arena = pooled.slot_arena; e->spare_slot_bytes = pooled.slot_bytes;
            e->d_cuts = pooled.d_cuts; e->h_cuts = pooled.h_cuts; e->cuts_bytes = pooled.cuts_bytes;
            e->d_out = pooled.d_out; e->h_out = pooled.h_out; e->out_bytes_cap = pooled.out_bytes;
            e->h_up = pooled.h_up; e->h_up_bytes = pooled.h_up_bytes; e->d_up = pooled.d_up; e->d_up_bytes = pooled.d_up_bytes;
            e->r_arena = pooled.r_arena; e->r_arena_bytes = pooled.r_arena_bytes;
            e->copy_stream = pooled.copy_stream; e->ev_group = pooled.ev_group;
        } else {
            HIPC(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
        }
        {   // snapshot, unrestricted flags and pivot trace: one allocation
            const size_t cells = (size_t)e->cap_rows * e->ld;
            for (int pass = 0; pass < 2; pass++) {
                Carver cv{pass ? e->static_arena : nullptr, 0};
                e->d_unr = cv.take<uint8_t>((size_t)e->n_idx);
                e->d_isint = cv.take<uint8_t>((size_t)e->n_idx);
                e->snap_A = cv.take<double>(cells);
                e->snap_rhs = cv.take<double>((size_t)e->cap_rows);
                e->snap_ldT = (e->cap_rows + 1) & ~1;
                e->snap_AT = wglds_bytes(e->ld, e->cap_rows) <= WGLDS_MAX_BYTES ? cv.take<double>((size_t)e->snap_ldT * e->W) : nullptr;
                e->snap_vibr = cv.take<int32_t>((size_t)e->cap_rows);
                e->snap_vibc = cv.take<int32_t>((size_t)e->W);
                e->snap_rbv = cv.take<int32_t>((size_t)e->n_idx);
                e->snap_cbv = cv.take<int32_t>((size_t)e->n_idx);
                e->s.trace = cv.take<int2>((size_t)TRACE_CAP);
                e->d_nnz = cv.take<unsigned long long>(1);
                e->d_done_count = cv.take<int>(1);
                if (!pass) {
                    e->static_bytes = cv.off + 256;
                    if (have && pooled.static_bytes >= e->static_bytes) {
                        e->static_arena = pooled.static_arena;
                        e->static_bytes = pooled.static_bytes;
                    } else {
                        if (have) hipFree(pooled.static_arena);
                        HIPC(hipMalloc(&e->static_arena, e->static_bytes));
                    }
                }
            }
            e->s.trace_cap = TRACE_CAP;
        }
        HIPC(hipMemsetAsync(e->d_unr, 0, e->n_idx, e->stream));
        HIPC(hipMemsetAsync(e->d_done_count, 0, sizeof(int), e->stream));
        int r = ensure_slots(e, 1);
        if (r) return r;
        if (!have) {
            HIPC(hipHostMalloc(&e->h_state, sizeof(DevState) + 64));
            memset(e->h_state, 0, sizeof(DevState) + 64);
            HIPC(hipEventCreate(&e->ev_begin));
            HIPC(hipEventCreate(&e->ev_end));
        }
        // (no synchronisation here: the two memsets above are ordered in front of everything this engine will ever enqueue -- one
        //  stream --, and a Solve() of a mid-size LP creates an engine per call)
        // the completion flag lives behind the pinned state and travels with it through the resource pool: continue the
        // previous owner's sequence (a fresh counter would meet the old owner's numbers again)
        e->done_seq = *reinterpret_cast<unsigned*>(reinterpret_cast<char*>(e->h_state) + sizeof(DevState));
        return JSLP_OK;
    };
    rc = init();
    if (rc) { jslp_engine_destroy(e); return rc; }
    *out = e;
    return JSLP_OK;
}

extern "C" void jslp_engine_destroy(jslp_engine* e) {
    if (!e) return;
    hipSetDevice(e->device);
    if (e->stream) hipStreamSynchronize(e->stream);
    if (e->copy_stream) hipStreamSynchronize(e->copy_stream);
    if (e->spare_slot_arena) hipFree(e->spare_slot_arena);
    bool parked = false;
    if (e->stream && e->ev_begin && e->ev_end && e->h_state && e->static_arena && e->slot_arena) {
        PooledRes r;
        r.device = e->device; r.stream = e->stream; r.ev_begin = e->ev_begin; r.ev_end = e->ev_end; r.h_state = e->h_state;
        r.static_arena = e->static_arena; r.static_bytes = e->static_bytes;
        r.slot_arena = e->slot_arena; r.slot_bytes = e->slot_bytes;
        r.d_cuts = e->d_cuts; r.h_cuts = e->h_cuts; r.cuts_bytes = e->cuts_bytes;
        r.d_out = e->d_out; r.h_out = e->h_out; r.out_bytes = e->out_bytes_cap;
        r.h_up = e->h_up; r.h_up_bytes = e->h_up_bytes; r.d_up = e->d_up; r.d_up_bytes = e->d_up_bytes;
        r.r_arena = e->r_arena; r.r_arena_bytes = e->r_arena_bytes;
        r.copy_stream = e->copy_stream; r.ev_group = e->ev_group;
        parked = pool_give(r);
    }
    free_slots(e, parked);
    if (parked) {
        e->static_arena = nullptr; e->h_state = nullptr; e->ev_begin = e->ev_end = nullptr; e->stream = nullptr;
        e->d_cuts = e->h_cuts = nullptr; e->d_out = e->h_out = nullptr; e->h_up = e->d_up = nullptr; e->r_arena = nullptr;
        e->copy_stream = nullptr; e->ev_group = nullptr;
    }
    if (e->h_up) hipHostFree(e->h_up);
    hipFree(e->d_up); hipFree(e->d_watch); hipFree(e->d_cnt); hipFree(e->r_arena);
    hipFree(e->static_arena); hipFree(e->snap_oo);
    drop_checkpoints(e, 1);
    hipFree(e->arena32);
    hipFree(e->f_uflags);
    hipFree(e->f_oo1);
    hipFree(e->f_buf1); hipFree(e->f_cands[0]); hipFree(e->f_cands[1]); hipFree(e->f_pcol[0]); hipFree(e->f_pcol[1]);
    hipFree(e->f_st[0]); hipFree(e->f_st[1]);
    hipFree(e->d_cuts); if (e->h_cuts) hipHostFree(e->h_cuts);
    hipFree(e->d_out); if (e->h_out) hipHostFree(e->h_out);
    if (e->h_state) hipHostFree(e->h_state);
    for (auto ev : e->ev_pool) hipEventDestroy(ev);
    if (e->ev_begin) hipEventDestroy(e->ev_begin);
    if (e->ev_end) hipEventDestroy(e->ev_end);
    if (e->ev_group) hipEventDestroy(e->ev_group);
    if (e->copy_stream) hipStreamDestroy(e->copy_stream);
    if (e->stream) hipStreamDestroy(e->stream);
    delete e;
}

// upload staging: pinned [matrix H0 x W | vibr | vibc | unrestricted list] and its device twin
static size_t up_matrix_bytes(const jslp_engine* e) { return ((sizeof(double) * (size_t)e->H0 * e->W) + 255) & ~(size_t)255; }
static size_t up_blob_bytes(const jslp_engine* e) { return sizeof(int32_t) * ((size_t)e->H0 + e->W + e->n_idx); }
static int ensure_up(jslp_engine* e) {
    const size_t total = up_matrix_bytes(e) + up_blob_bytes(e);
    if (e->h_up_bytes < total) {
        if (e->h_up) hipHostFree(e->h_up);
        e->h_up = nullptr; e->h_up_bytes = 0;
        HIPC(hipHostMalloc(&e->h_up, total));
        e->h_up_bytes = total;
    }
    if (e->d_up_bytes < total) {
        hipFree(e->d_up);
        e->d_up = nullptr; e->d_up_bytes = 0;
        HIPC(hipMalloc(&e->d_up, total));
        e->d_up_bytes = total;
    }
    return JSLP_OK;
}

extern "C" int jslp_engine_host_matrix(jslp_engine* e, double** matrix, int64_t* n_doubles) {
    if (!e || !matrix) return fail(JSLP_ERR_ARG, "host_matrix: null pointer");
    HIPC(hipSetDevice(e->device));
    int rc = ensure_up(e);
    if (rc) return rc;
    memset(e->h_up, 0, sizeof(double) * (size_t)e->H0 * e->W);  // a fresh Float64Array is zero-filled (tableau.ts:304)
    e->host_matrix_out = 1;
    *matrix = reinterpret_cast<double*>(e->h_up);
    if (n_doubles) *n_doubles = (int64_t)e->H0 * e->W;
    return JSLP_OK;
}

extern "C" int jslp_engine_upload(jslp_engine* e, const double* matrix, const int32_t* var_index_by_row,
                                  const int32_t* var_index_by_col, const int32_t* unrestricted_var_indexes,
                                  int32_t n_unrestricted) {
    if (!e || !matrix || !var_index_by_row || !var_index_by_col) return fail(JSLP_ERR_ARG, "upload: null pointer");
    if (n_unrestricted < 0 || (n_unrestricted > 0 && !unrestricted_var_indexes)) return fail(JSLP_ERR_ARG, "upload: bad unrestricted list");
    if (n_unrestricted > e->n_idx) return fail(JSLP_ERR_ARG, "upload: more unrestricted variables than variable indexes");
    HIPC(hipSetDevice(e->device));
    const int32_t H = e->H0, W = e->W;
    int rc = ensure_up(e);
    if (rc) return rc;
    // everything crosses PCIe from pinned memory: the matrix as ONE DMA (straight from the buffer the host built it in when
    // it used jslp_engine_host_matrix), the maps as one small blob; the inverse maps, flags and state are built on the device
    const size_t mat = sizeof(double) * (size_t)H * W, mat_pad = up_matrix_bytes(e);
    int32_t* b_vibr = reinterpret_cast<int32_t*>(e->h_up + mat_pad);
    int32_t* b_vibc = b_vibr + H;
    int32_t* b_unr = b_vibc + W;
    b_vibr[0] = -1; b_vibc[0] = -1;
    int32_t max_idx = -1;
    for (int32_t r = 1; r < H; r++) {
        const int32_t v = var_index_by_row[r];
        if (v < 0 || v >= e->n_idx) return fail(JSLP_ERR_ARG, "upload: row variable index out of range");
        b_vibr[r] = v;
        max_idx = std::max(max_idx, v);
    }
    for (int32_t c = 1; c < W; c++) {
        const int32_t v = var_index_by_col[c];
        if (v < 0 || v >= e->n_idx) return fail(JSLP_ERR_ARG, "upload: column variable index out of range");
        b_vibc[c] = v;
        max_idx = std::max(max_idx, v);
    }
    e->max_uploaded_idx = max_idx;
    for (int32_t i = 0; i < n_unrestricted; i++) {
        const int32_t v = unrestricted_var_indexes[i];
        if (v < 0 || v >= e->n_idx) return fail(JSLP_ERR_ARG, "upload: unrestricted variable index out of range");
        b_unr[i] = v;
    }
    if (matrix != reinterpret_cast<const double*>(e->h_up)) memcpy(e->h_up, matrix, mat);  // pageable caller memory: stage it once
    hipStream_t s = e->stream;
    const size_t blob = sizeof(int32_t) * ((size_t)H + W + (size_t)n_unrestricted);
    char* d_blob = e->d_up + mat_pad;
    if (W == e->ld) {
        HIPC(hipMemcpyAsync(e->s.A, e->h_up, mat, hipMemcpyHostToDevice, s));
        HIPC(hipMemcpyAsync(d_blob, e->h_up + mat_pad, blob, hipMemcpyHostToDevice, s));
    } else {  // one DMA for matrix + blob, then the W -> ld repack on the device (padding columns become 0)
        HIPC(hipMemcpyAsync(e->d_up, e->h_up, mat_pad + blob, hipMemcpyHostToDevice, s));
        const long long cells = (long long)H * e->ld;
        hipLaunchKernelGGL(k_repack, dim3((unsigned)std::min<long long>(2048, (cells + 255) / 256)), dim3(256), 0, s, e->s.A,
                           reinterpret_cast<const double*>(e->d_up), (int)H, (int)W, (int)e->ld);
    }
    HIPC(hipMemsetAsync(e->d_nnz, 0, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(k_count_nnz, dim3(256), dim3(256), 0, s, e->s.A, (int)H, (int)e->ld, e->d_nnz);
    UploadBlob ub;
    ub.vibr = reinterpret_cast<const int32_t*>(d_blob);
    ub.vibc = ub.vibr + H;
    ub.unr = ub.vibc + W;
    ub.H = H; ub.n_unr = n_unrestricted; ub.n_idx = e->n_idx; ub.cap_rows = e->cap_rows;
    hipLaunchKernelGGL(k_upload_finish, dim3(1), dim3(1024), 0, s, e->s, ub, e->d_unr, e->d_isint);
    HIPC(hipGetLastError());
    HIPC(hipMemcpyAsync(e->h_state, e->d_nnz, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIPC(hipStreamSynchronize(s));  // the staging buffer (and the host's view of it) is free again
    e->nnz = (long long)*reinterpret_cast<unsigned long long*>(e->h_state);
    e->uploaded = 1;
    e->has_save = 0;
    e->root_seq += 1;
    e->slot0_synced = 0;
    e->slots_synced = 0;
    drop_checkpoints(e, 0);
    e->evaluation = 0;
    e->n_unr = n_unrestricted;
    e->s.has_unr = n_unrestricted > 0 ? 1 : 0;
    if (e->n_opt > 0) {  // a new model: optional objectives are set again by the caller
        hipFree(e->s.oo); hipFree(e->snap_oo);
        e->s.oo = nullptr; e->snap_oo = nullptr; e->s.n_opt = 0; e->s.oo_stride = 0; e->n_opt = 0;
    }
    return JSLP_OK;
}

extern "C" int jslp_engine_set_optional_objectives(jslp_engine* e, int32_t n, const double* rows) {
    if (!e || !e->uploaded) return fail(JSLP_ERR_STATE, "set_optional_objectives before upload");
    if (n < 0 || (n > 0 && !rows)) return fail(JSLP_ERR_ARG, "set_optional_objectives: bad arguments");
    HIPC(hipSetDevice(e->device));
    HIPC(hipStreamSynchronize(e->stream));
    hipFree(e->s.oo); hipFree(e->snap_oo);
    e->s.oo = nullptr; e->snap_oo = nullptr;
    e->n_opt = n;
    e->s.n_opt = n;
    e->s.oo_stride = (long long)n * e->ld;
    if (n > 0) {
        const size_t per = (size_t)e->s.oo_stride;
        HIPC(hipMalloc(&e->s.oo, sizeof(double) * per * std::max(1, e->n_slots)));
        HIPC(hipMalloc(&e->snap_oo, sizeof(double) * per));
        HIPC(hipMemset(e->s.oo, 0, sizeof(double) * per * std::max(1, e->n_slots)));
        HIPC(hipMemcpy2D(e->s.oo, sizeof(double) * e->ld, rows, sizeof(double) * e->W, sizeof(double) * e->W, n,
                         hipMemcpyHostToDevice));
    }
    return JSLP_OK;
}

extern "C" int jslp_engine_get_optional_objectives(jslp_engine* e, double* rows, int32_t* n_out) {
    if (!e || !e->uploaded) return fail(JSLP_ERR_STATE, "get_optional_objectives before upload");
    HIPC(hipSetDevice(e->device));
    if (n_out) *n_out = e->n_opt;
    if (rows && e->n_opt > 0) {
        HIPC(hipStreamSynchronize(e->stream));
        HIPC(hipMemcpy2D(rows, sizeof(double) * e->W, e->s.oo, sizeof(double) * e->ld, sizeof(double) * e->W, e->n_opt,
                         hipMemcpyDeviceToHost));
    }
    return JSLP_OK;
}

static Ctx host_ctx(const jslp_engine* e, int check_cycles) {
    Ctx c{};
    c.A = e->s.A; c.vibr = e->s.vibr; c.vibc = e->s.vibc; c.rbv = e->s.rbv; c.cbv = e->s.cbv; c.unr = e->s.unr;
    c.prow = e->s.prow; c.pcol = e->s.pcol; c.dirty = e->s.dirty; c.rhs = nullptr; c.oo = e->s.oo; c.n_opt = e->n_opt; c.st = e->s.st; c.hist = e->s.hist; c.hist_cap = e->s.hist_cap;
    c.trace = e->s.trace; c.trace_cap = e->s.trace_cap; c.ld = e->ld; c.W = e->W; c.check_cycles = check_cycles;
    c.batch = e->batch; c.use_partial = e->use_partial; c.precision = e->precision; c.stop_at_phase2 = 0;
    c.has_unr = e->n_unr > 0 ? 1 : 0;
    c.cnt = e->s.cnt;
    return c;
}

static int iters_cap(const jslp_engine* e) {
    // the reference has no iteration limit; this cap only turns an endless cycle (cycle check disabled) into
    // an error instead of a hung GPU
    long long cap = 2000000LL + 200LL * ((long long)e->cap_rows + e->W);
    return (int)std::min<long long>(cap, 2000000000LL);
}

// dynamic LDS of the LDS-resident one-workgroup kernels (jslp_wglds.hip.h); 0 = use the generic kernels
static size_t wglds_smem(const jslp_engine* e) {
    if (!e->use_wglds) return 0;  // (optional objectives: round 3 -- their rows stay in the slot's global copy, jslp_wglds.hip.h)
    const size_t b = wglds_bytes(e->ld, e->cap_rows);
    return b <= WGLDS_MAX_BYTES ? b : 0;
}

// one workgroup for the whole simplex(): tiny tableaus, and SPARSE ones up to a few million cells -- a pivot of the
// LDS-resident one-workgroup kernel touches only the rows and columns the reference's gates let through (Monster LP, 1 %
// dense: ~5 us per pivot against ~12 for the chip-wide register-resident kernel, whose hand-off latency does not shrink with
// the work); dense tableaus of that size belong to the chip
static bool xl_fits(const jslp_engine* e, int H);
static const long long WG_CELLS_SPARSE = 1024LL * 1024;  // (Vendor Selection, 2.8 M cells, 0.3 % dense: 38.7 ms in one workgroup against 9.0 ms register-resident -- fill-in makes its pivots touch hundreds of rows)
static const double WG_SPARSE_DENSITY = 0.05;
static bool use_wg_single(const jslp_engine* e) {
    if (e->force_path == 1) return true;
    if (e->force_path == 2 || e->force_path == 3) return false;
    const long long cells = (long long)e->cap_rows * e->ld;
    if (e->force_xl) return false;
    if (cells <= WG_CELLS_SINGLE) return true;
    // (JSLP_XL=1: the XCD-local resident geometry takes the sparse mid-size LPs too)
    if (xl_fits(e, e->cap_rows) && !e->no_resident && e->force_path == 0) return false;
    return wglds_smem(e) != 0 && cells <= WG_CELLS_SPARSE && e->nnz >= 0 &&
           (double)e->nnz <= WG_SPARSE_DENSITY * (double)e->H0 * (double)e->W;
}

#define JSLP_F_MAXNT 4
static int fused_wide_on() {  // JSLP_FUSED_WIDE=0: tableaus wider than 4096 columns through k_select + k_update, as before round 5
    static int v = -1;
    if (v < 0) { const char* t = getenv("JSLP_FUSED_WIDE"); v = (t && t[0] == '0') ? 0 : 1; }
    return v;
}
// the fused one-launch-per-pivot phase 2 (see k_pivot_fused for the preconditions)
static bool fused_eligible(const jslp_engine* e) {
    if (e->force_path == 2) return false;
    // (round 5: three and four column tiles per lane -- 4096 < ld <= 8192 -- for tableaus without unrestricted variables / optional objectives:
    //  3001 x 5001 ran k_update at 0.83 of the HBM peak but, with the one-workgroup k_select in front of every pivot, 17.3 k pivots/s = 0.52
    //  end to end, profiles/r05_streaming_workload_rows.md)
    const int max_tiles = (e->n_unr == 0 && e->n_opt == 0 && fused_wide_on()) ? JSLP_F_MAXNT : 2;
    return e->ld <= max_tiles * JSLP_F_TW && e->cap_rows <= 64 * JSLP_F_MAXG && e->precision >= 1e-15;
}

// Geometry of the register-resident kernel for this tableau: lanes x columns per lane must cover a row (ld), rows per
// workgroup x 256 workgroups the height, and rows x columns per lane must fit the lane's registers.  0 = does not fit.
//   1: <1024, 2, 8>   ld <= 2048, H <= 2048 (the headline shape)      2: <512, 4, 8>  (JSLP_RES_CPT=4, measured slower)
//   3: <512, 4, 16>   ld <= 2048, H <= 4096                            4: <512, 6, 12> ld <= 3072, H <= 3072 (3001 x 3001: 72 MB)
//   5: <512, 8, 8>    ld <= 4096, H <= 2048
//   6: <512, 2, 32> XL ld <= 1024, H <= 1024, on the <= 32 workgroups of ONE XCD (round 4): the hand-offs of a pivot go through
//      that XCD's L2 instead of memory.  Mid-size tableaus: Monster LP 625 x 553, Monster_II's root 945 x 925, 501 x 501 ...
// (256-lane geometries -- ONE wave per SIMD, 512 registers per lane: <256, 8, 8> compiles without a spill -- were measured and
//  dropped: 72.5 k against 105.7 k pivots/s on a 2001 x 2001 LP, 17.4 k on 4001 x 2001, r02_z: a lone wave per SIMD does not hide
//  its own instruction latency)
// Round 5: the XCD-local instances are compiled into the TEST library only (-DJSLP_CHAOS_BUILD implies -DJSLP_WITH_XL; a hipcc run with
// -DJSLP_WITH_XL builds a product-like library with them): the default policy never picked the geometry (parity with the chip-wide kernel on
// an eighth of the chip, profiles/r04_xl_times.md), and its two instances cost the shipped library 40 s of compile time and 320 spilled
// SGPRs of dead weight.  The shipped library ignores JSLP_XL and refuses JSLP_FORCE_PATH=xl loudly (jslp_engine_create).
static bool xl_fits(const jslp_engine* e, int H) {
#ifndef JSLP_WITH_XL
    (void)e; (void)H;
    return false;
#else
    return (e->xl_on || e->force_xl) && e->n_unr == 0 && e->n_opt == 0 && e->ld <= 1024 && H <= JSLP_XL_MAXG * JSLP_R_MAXROWS && e->precision >= 1e-15 &&
           !(getenv("JSLP_RES_LEAN") && atoi(getenv("JSLP_RES_LEAN")) == 0);
#endif
}
// variable indexes that can be live in a tableau of height H: constraints and variables are numbered first, cut slacks continue from
// lastElementIndex = width + height - 2 (tableau.ts:312-316) -- e->n_idx is the CAPACITY (every cut row the engine has room for)
// (an uploaded tableau need not follow the reference's contiguous numbering -- Model index reuse / removal, a direct C-API caller --
//  so the largest index upload() saw in the two maps counts too: the lean kernels' LDS copy of the "unrestricted" flags is indexed with it)
static int live_index_bound(const jslp_engine* e, int H) { return std::min<int>(e->n_idx, std::max<int>(e->W + H + 2, e->max_uploaded_idx + 1)); }
#if defined(JSLP_SPLIT_TU)
extern "C" __attribute__((visibility("hidden"))) int jslpx_resident_launch_1(const int* key, unsigned grid, const void* rc, void* stream);
extern "C" __attribute__((visibility("hidden"))) int jslpx_resident_launch_2(const int* key, unsigned grid, const void* rc, void* stream);
#endif
static int resident_geometry(const jslp_engine* e, int H) {
    if (e->no_resident || e->force_path == 2 || e->precision < 1e-15) return 0;  // (see k_pivot_fused for the precision condition)
    const int rpb = (H + JSLP_F_MAXG - 1) / JSLP_F_MAXG;
    // (JSLP_FORCE_PATH=resident keeps meaning the chip-wide geometries: the tests that force them at small sizes must keep covering them)
    if (xl_fits(e, H) && !getenv("JSLP_RES_GEOM") && (e->force_xl || !e->force_resident)) return 6;
    if (e->n_opt > 0) {
        // optional objectives: the lean build of the headline geometry keeps up to three rows of them in registers (round 3);
        // everything else (unrestricted variables, taller / wider tableaus, more rows) runs them through the fused pipeline
        const bool lean_ok = !(getenv("JSLP_RES_LEAN") && atoi(getenv("JSLP_RES_LEAN")) == 0) && e->n_unr == 0;
        if (!lean_ok || e->n_opt > 3) return 0;
        if (e->ld <= 2048 && rpb <= 8) return 1;
        // round 4: the tall geometry too (its lean phase 2 has the registers for three more rows per lane: 512 lanes x 4 columns);
        // the 6- and 8-column geometries do not (the objective rows would spill): the fused pipeline keeps those shapes
        const char* wt = getenv("JSLP_RES_WIDE_TALL");
        if (wt && atoi(wt) == 0 && !e->force_resident) return 0;
        return (e->ld <= 2048 && rpb <= 16) ? 3 : 0;
    }
    if (const char* gx = getenv("JSLP_RES_GEOM")) {  // experiments: force a geometry the tableau fits (lean build, no unrestricted variables)
        static const int rows_of[6] = {0, 8, 8, 16, 12, 8}, ld_of[6] = {0, 2048, 2048, 2048, 3072, 4096};
        const int g = atoi(gx);
        if (g >= 1 && g <= 5 && e->n_unr == 0 && e->ld <= ld_of[g] && rpb <= rows_of[g]) return g;
    }
    if (e->ld <= 2048 && rpb <= 8) return e->res_cpt == 4 ? 2 : 1;
    // The taller / wider geometries hold 64-72 MB of tableau in the 128 MB of vector registers and spill ~0.5 KB per lane to
    // scratch (560+ scratch loads in the pivot loop): measured at the end of round 2 they LOSE to the streaming kernels --
    // 4001 x 2001: 22.3 k pivots/s against 36.4 k through k_pivot_fused; 2501 x 2001: 26.6 k against 48.8 k; 3001 x 3001: 21.5 k
    // against 23.7 k through k_select + k_update; 2001 x 4001: 20.1 k against 26.1 k (tools/tall_one.py, r02_z).  They stay
    // built and tested (JSLP_FORCE_PATH=resident, JSLP_RES_WIDE_TALL=1) but the default policy no longer picks them.
    // Round 3: the LEAN build's pipelined phase 2 does fit (jslp_resident_pipe.hip.h; 4 scratch loads per pivot) and beats the
    // streaming kernels 2-3x on these shapes (r03_j), so tableaus it can take -- no unrestricted variables -- get these geometries
    // by default for their phase 2 (phase 1 through the fused pipeline: run_simplex).  JSLP_RES_WIDE_TALL=0 switches that off, =1
    // also sends the general build's (unrestricted variables) there.
    const char* wt_env = getenv("JSLP_RES_WIDE_TALL");  // (read per call: tests switch it inside one process)
    const int wide_tall = wt_env ? atoi(wt_env) : -1;
    const bool lean_env = !(getenv("JSLP_RES_LEAN") && atoi(getenv("JSLP_RES_LEAN")) == 0);
    // (round 4: the lean build takes unrestricted variables too -- per-lane column masks, the flags of all variable indexes in LDS:
    //  n_idx <= JSLP_R_LUNR; the tall / wide geometries have NO general build any more -- it spilled ~0.5 KB per lane and lost to the
    //  streaming kernels -- so what the lean build cannot take goes to the fused pipeline)
    const bool lean = lean_env && (e->n_unr == 0 || live_index_bound(e, H) <= JSLP_R_LUNR);
    if (!lean) return 0;
    if (!e->force_resident && wide_tall == 0) return 0;
    if (e->ld <= 2048 && rpb <= 16) return 3;
    if (e->ld <= 3072 && rpb <= 12) return 4;
    if (e->ld <= 4096 && rpb <= 8) return 5;
    return 0;
}
static bool resident_eligible(const jslp_engine* e, int H) { return resident_geometry(e, H) != 0; }

static int ensure_resident(jslp_engine* e, bool want_hist = false) {
    // (`want_hist`: the lean kernel's per-workgroup copies of the cycle-check history, 256 MiB -- carved only once a solve with the
    //  cycle check on takes the lean build: pool members, short-lived engines and check-off solves never pay for it)
    if (want_hist) e->r_want_hist = 1;
    //  (JSLP_RES_GEOM can force a geometry WIDER than the tableau needs -- 6 or 8 columns per lane on ld <= 2048 / <= 3072 --, whose slots are
    //   lanes x columns per lane doubles whatever ld is: with the knob set every slot is sized for the widest geometry, 4096 doubles + skew;
    //   the knob is read per call, so an engine carved without it is carved again)
    const size_t slot_doubles = getenv("JSLP_RES_GEOM") ? (size_t)4096 + JSLP_PUB_SKEW / 8
                                : (e->ld <= 2048 ? (size_t)e->ld : ((size_t)e->ld + 1023) / 1024 * 1024 + JSLP_PUB_SKEW / 8);
    if (e->r_sync && (!e->r_want_hist || e->r_hist_all) && e->r_slot_doubles >= slot_doubles) return JSLP_OK;
    if (e->r_sync) HIPC(hipStreamSynchronize(e->stream));  // re-carving: nothing in flight may still use the old layout
    // hand-off buffers and the safety-net copy of slot 0 (matrix, maps, state) carved from ONE allocation, which the
    // resource pool hands from engine to engine (hipMalloc / hipFree of these cost a small Solve more than its pivots)
    for (int pass = 0; pass < 2; pass++) {
        Carver cv{pass ? e->r_arena : nullptr, 0};
        e->r_gran = cv.take<u64_t>(JSLP_R_SYNC_WORDS);
        // (slots ld doubles apart; the 6- and 8-column geometries' permuted layout -- jslp_resident_pipe.hip.h, SLOT -- takes lanes x columns per lane
        //  = 3072 / 4096 doubles plus a skew per slot)
        for (int i = 0; i < 2; i++) e->r_rows[i] = cv.take<u64_t>((size_t)JSLP_F_MAXG * slot_doubles);
        e->r_hist_all = e->r_want_hist ? cv.take<int2>((size_t)JSLP_F_MAXG * JSLP_PIPE_GHIST) : nullptr;  // every workgroup's own copy of the cycle-check history (lean kernel)
        e->r_sync = cv.take<unsigned>(16);
        e->rb_A = cv.take<double>((size_t)e->cap_rows * e->ld);
        e->rb_vibr = cv.take<int32_t>((size_t)e->cap_rows);
        e->rb_vibc = cv.take<int32_t>((size_t)e->W);
        e->rb_rbv = cv.take<int32_t>((size_t)e->n_idx);
        e->rb_cbv = cv.take<int32_t>((size_t)e->n_idx);
        e->r_backup_st = cv.take<DevState>(1);
        e->r_ctx_dev = reinterpret_cast<Ctx*>(cv.take<char>(sizeof(Ctx) + 16));  // (+ the address of the host-abort word: JSLP_HOST_ABORT_CHECK)
        e->r_slot_doubles = slot_doubles;
        if (!pass && e->r_arena_bytes < cv.off + 256) {
            hipFree(e->r_arena);
            e->r_arena = nullptr; e->r_arena_bytes = 0; e->r_sync = nullptr;
            HIPC(hipMalloc(&e->r_arena, cv.off + 256));
            e->r_arena_bytes = cv.off + 256;
        }
    }
    return JSLP_OK;
}

static int ensure_fused(jslp_engine* e) {
    if (e->f_buf1) return JSLP_OK;
    const size_t cells = (size_t)e->cap_rows * e->ld;
    HIPC(hipMalloc(&e->f_buf1, sizeof(double) * cells));
    HIPC(hipMemsetAsync(e->f_buf1, 0, sizeof(double) * cells, e->stream));
    for (int i = 0; i < 2; i++) {
        HIPC(hipMalloc(&e->f_cands[i], sizeof(FCand) * JSLP_F_MAXG));
        HIPC(hipMalloc(&e->f_pcol[i], sizeof(double) * e->cap_rows));
        HIPC(hipMalloc(&e->f_st[i], sizeof(DevState)));
        HIPC(hipMemsetAsync(e->f_st[i], 0, sizeof(DevState), e->stream));
    }
    HIPC(hipMalloc(&e->f_uflags, 2 * ((size_t)e->ld + (size_t)e->cap_rows)));
    HIPC(hipMemsetAsync(e->f_uflags, 0, 2 * ((size_t)e->ld + (size_t)e->cap_rows), e->stream));
    return JSLP_OK;
}

static int ensure_fused_oo(jslp_engine* e) {  // (the number of optional objectives is set per model, after create())
    if (e->n_opt <= e->f_oo1_rows) return JSLP_OK;
    HIPC(hipStreamSynchronize(e->stream));
    hipFree(e->f_oo1);
    e->f_oo1 = nullptr; e->f_oo1_rows = 0;
    HIPC(hipMalloc(&e->f_oo1, sizeof(double) * (size_t)e->n_opt * e->ld));
    HIPC(hipMemsetAsync(e->f_oo1, 0, sizeof(double) * (size_t)e->n_opt * e->ld, e->stream));
    e->f_oo1_rows = e->n_opt;
    return JSLP_OK;
}
static FusedCtx make_fused_ctx(const jslp_engine* e, const Ctx& c, int H) {
    FusedCtx f;
    f.c = c;
    f.buf[0] = e->s.A; f.buf[1] = e->f_buf1;
    for (int i = 0; i < 2; i++) { f.cands[i] = e->f_cands[i]; f.pcol[i] = e->f_pcol[i]; f.fst[i] = e->f_st[i]; }
    f.rpb = (H + JSLP_F_MAXG - 1) / JSLP_F_MAXG;
    f.G = (H + f.rpb - 1) / f.rpb;
    f.H = H;
    f.nt = e->nt;
    for (int i = 0; i < 2; i++) {
        f.ucol[i] = e->f_uflags + (size_t)i * ((size_t)e->ld + e->cap_rows);
        f.urow[i] = f.ucol[i] + e->ld;
    }
    f.oo[0] = e->s.oo; f.oo[1] = e->f_oo1;
    return f;
}
static int fused_p1_on() {  // phase 1 of the large LPs through k_fused_p1 (one launch per pivot) instead of k_select + k_update
    static int v = -1;
    if (v < 0) {
        const char* t = getenv("JSLP_FUSED_P1");  // tuning knob
        v = t ? atoi(t) : 1;
    }
    return v;
}

static dim3 update_grid(const jslp_engine* e, int H) {
    return dim3((e->ld + JSLP_UPD_COLS - 1) / JSLP_UPD_COLS, (H + JSLP_UPD_ROWS - 1) / JSLP_UPD_ROWS, 1);
}

// Math.round: nearest integer, ties toward +Infinity
static double js_round(double x) {
    if (!std::isfinite(x)) return x;
    const double f = floor(x);
    return (x - f >= 0.5) ? f + 1.0 : f;
}

// checkForCycles (simplex.ts:415-440) on the downloaded history: only to rebuild the exact [start, length]
// the reference pushes to model.messages; hit / no-hit was decided on the device.
static void cycle_message(const std::vector<int2>& h, int32_t* start, int32_t* len) {
    const long long n = (long long)h.size();
    for (long long e1 = 0; e1 < n - 1; e1++)
        for (long long e2 = e1 + 1; e2 < n; e2++)
            if (h[e1].x == h[e2].x && h[e1].y == h[e2].y) {
                if (e2 - e1 > n - e2) break;
                bool found = true;
                for (long long i = 1; i < e2 - e1; i++)
                    if (h[e1 + i].x != h[e2 + i].x || h[e1 + i].y != h[e2 + i].y) { found = false; break; }
                if (found) { *start = (int32_t)e1; *len = (int32_t)(e2 - e1); return; }
            }
    *start = -1; *len = 0;
}

static int state_error(const DevState& st) {
    switch (st.err) {
        case ERR_NONE: return JSLP_OK;
        case ERR_HIST_FULL: return fail(JSLP_ERR_CAPACITY, "simplex: cycle-check history capacity exceeded");
        case ERR_ITER_LIMIT: return fail(JSLP_ERR_CAPACITY, "simplex: iteration safety cap reached (endless cycle?)");
        case ERR_CUT_ARG: return fail(JSLP_ERR_ARG, "add_cuts: variable index out of range or neither basic nor non-basic");
        case ERR_CAPACITY: return fail(JSLP_ERR_CAPACITY, "add_cuts: row / element-index capacity exceeded");
        case ERR_BARRIER: return fail(JSLP_ERR_DEVICE, "resident simplex kernel: grid barrier timed out (workgroups not co-resident?)");
        case ERR_NOT_SYNCED: return fail(JSLP_ERR_STATE, "internal: one-launch node kernel used on a slot that is not in sync with the snapshot");
    }
    return fail(JSLP_ERR_DEVICE, "unknown device error code");
}

// fold a finished DevState into the ABI result (evaluation semantics: see jslp_simplex_result)
static int fill_result(jslp_engine* e, const DevState& st, int slot, double prev_evaluation, jslp_simplex_result* out,
                       double* evaluation_out) {
    memset(out, 0, sizeof *out);
    out->feasible = st.feasible;
    out->bounded = st.bounded;
    out->optimal = st.optimal;
    out->unbounded_var_index = st.bounded ? -1 : st.unbounded_var;
    out->pivots_phase1 = st.it1;
    out->pivots_phase2 = st.entered_phase2 ? st.it2 : -1;
    out->cycle_phase = st.cycle_phase;
    out->height = st.H;
    out->obj_cell = st.obj_cell;
    double ev = prev_evaluation;
    if (st.optimal) {  // setEvaluation (tableau.ts:420-426)
        const double rc = js_round(1.0 / e->precision);
        ev = js_round((2.220446049250313e-16 + st.obj_cell) * rc) / rc;
    } else if (!st.bounded) {
        ev = -INFINITY;
    }
    out->evaluation = ev;
    if (evaluation_out) *evaluation_out = ev;
    if (st.cycle_phase && st.hist_n > 0) {
        std::vector<int2> h(st.hist_n);
        if (hipMemcpy(h.data(), e->s.hist + (size_t)slot * e->s.hist_cap, sizeof(int2) * st.hist_n, hipMemcpyDeviceToHost) != hipSuccess)
            return fail(JSLP_ERR_DEVICE, "cycle history read-back failed");
        cycle_message(h, &out->cycle_start, &out->cycle_length);
    }
    return JSLP_OK;
}

static int ensure_events(jslp_engine* e, size_t n) {
    while (e->ev_pool.size() < n) {
        hipEvent_t ev;
        HIPC(hipEventCreate(&ev));
        e->ev_pool.push_back(ev);
    }
    return JSLP_OK;
}

// host half of the work counters: one finished simplex() (a relaxation when it came with restore + cuts)
static void account(jslp_engine* e, const DevState& st, int is_relaxation) {
    if (!e->counting) return;
    e->wc.simplex_calls += 1;
    e->wc.relaxations += is_relaxation ? 1 : 0;
    e->wc.pivots += (long long)st.it1 + st.it2;
    e->wc.height_sum += st.H;
}

// simplex() of the live tableau (slot 0), leaving the final DevState in e->h_state
static int run_simplex(jslp_engine* e, int check_cycles) {
    hipStream_t s = e->stream;
    e->slot0_synced = 0;  // the chip-wide kernels do not maintain the dirty-row flags (k_begin zeroes st.gen)
    e->abort_injected = 0;
    const int cap = iters_cap(e);
    HIPC(hipEventRecord(e->ev_begin, s));
    if (use_wg_single(e)) {
        e->last_path = "workgroup";
        if (const size_t lds = wglds_smem(e); lds && e->s.n_opt > 0)  // optional objectives: a build of their own (jslp_wglds.hip.h)
            hipLaunchKernelGGL((k_simplex_lds<JSLP_WG_THREADS, true>), dim3(1), dim3(JSLP_WG_THREADS), lds, s, e->s, 0, check_cycles, cap, (int)e->cap_rows);
        else if (lds)
            hipLaunchKernelGGL((k_simplex_lds<JSLP_WG_THREADS>), dim3(1), dim3(JSLP_WG_THREADS), lds, s, e->s, 0, check_cycles, cap, (int)e->cap_rows);
        else
            hipLaunchKernelGGL((k_simplex_wg<JSLP_WG_THREADS, 4096>), dim3(1), dim3(JSLP_WG_THREADS), 0, s, e->s, 0, check_cycles, cap);
        HIPC(hipGetLastError());
        HIPC(hipMemcpyAsync(e->h_state, e->s.st, sizeof(DevState), hipMemcpyDeviceToHost, s));
        HIPC(hipEventRecord(e->ev_end, s));
        HIPC(hipStreamSynchronize(s));
    } else {
        Ctx c = host_ctx(e, check_cycles);
        const bool fused = fused_eligible(e);
        e->last_path = "select+update";
        c.stop_at_phase2 = fused ? 1 : 0;
        // height is fixed during a simplex call; read it once
        HIPC(hipMemcpyAsync(e->h_state, e->s.st, sizeof(DevState), hipMemcpyDeviceToHost, s));
        HIPC(hipStreamSynchronize(s));
        const int H = e->h_state->H;
        const dim3 grid = update_grid(e, H);
        hipLaunchKernelGGL(k_begin, dim3(1), dim3(1), 0, s, e->s, 0, cap);
        // ---- register-resident path: the WHOLE simplex (phase 1 and phase 2) in one cooperative launch -------------
        bool resident_done = false;
        bool handed_to_streaming = false;  // the lean kernel left mid-solve and the general build cannot continue (optional objectives)
        const int geometry = resident_geometry(e, H);
        // `it_before`: pivots the solve had done when the kernel is launched (0, or a phase 1 done by the fused pipeline)
        auto run_resident = [&](long long it_before) -> int {
            const bool lean_on = !(getenv("JSLP_RES_LEAN") && atoi(getenv("JSLP_RES_LEAN")) == 0);
            const bool lean_unr_ok = e->n_unr == 0 || (live_index_bound(e, H) <= JSLP_R_LUNR && e->n_opt == 0);  // (the lean build's LDS copy of the "unrestricted" flags)
            int r = ensure_resident(e, lean_on && lean_unr_ok && check_cycles != 0);
            if (r) return r;
            ResCtx rc;
            rc.c = c;
            for (int i = 0; i < 2; i++) {
                rc.gran[i] = e->r_gran + (size_t)i * JSLP_F_MAXG * JSLP_R_GRAN;
                rc.rowflag[i] = e->r_gran + (size_t)2 * JSLP_F_MAXG * JSLP_R_GRAN + (size_t)i * JSLP_F_MAXG;
                rc.rows_pub[i] = e->r_rows[i];
            }
            rc.decision[0] = e->r_gran + (size_t)2 * JSLP_F_MAXG * (JSLP_R_GRAN + 1);
            rc.decision[1] = rc.decision[0] + 8;
            rc.verdict[0] = rc.decision[0] + 16;
            rc.verdict[1] = rc.decision[0] + 24;
            rc.gor[0] = rc.decision[0] + 32;
            rc.gor[1] = rc.gor[0] + JSLP_F_MAXG;
            rc.gran16 = e->r_gran + JSLP_R_SYNC_WORDS_GENERAL;  // [2][MAXG] granules, 64 bytes apart
            rc.hist_all = e->r_hist_all;
            for (int i = 0; i < 2; i++) rc.rowflagc[i] = e->r_gran + JSLP_R_SYNC_WORDS_GENERAL + 2 * JSLP_F_MAXG * 8 + (size_t)i * JSLP_R_FLAGCOPIES * JSLP_F_MAXG;
            rc.abort_flag = e->r_sync + 4;
            HIPC(hipMemsetAsync(e->r_gran, 0, sizeof(u64_t) * JSLP_R_SYNC_WORDS, s));  // tags restart at 1
            rc.census = e->r_gran + JSLP_R_SYNC_WORDS - JSLP_F_MAXG;
            rc.cdev = e->r_ctx_dev;
            HIPC(hipMemcpyAsync(e->r_ctx_dev, &rc.c, sizeof(Ctx), hipMemcpyHostToDevice, s));  // (200 bytes; the kernel's one committing thread reads the map / trace pointers from it)
            {   // the host-abort word (pinned, behind the state and the retry counter) and its address behind the device copy of the context
                unsigned* const h_abort = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(e->h_state) + sizeof(DevState) + 16);
                *h_abort = 0u;
                e->h_abort_ptr = h_abort;
                HIPC(hipMemcpyAsync(reinterpret_cast<char*>(e->r_ctx_dev) + sizeof(Ctx), &e->h_abort_ptr, sizeof(unsigned*), hipMemcpyHostToDevice, s));
            }
            rc.rpb = geometry == 6 ? (H + JSLP_XL_MAXG - 1) / JSLP_XL_MAXG : (H + JSLP_F_MAXG - 1) / JSLP_F_MAXG;
            if (const char* rx = getenv("JSLP_RES_RPB"); rx && geometry != 6) {  // experiments: more rows per workgroup = fewer workgroups (<= the geometry's rows)
                static const int rows_of[6] = {0, 8, 8, 16, 12, 8};
                const int want = atoi(rx);
                if (want > rc.rpb && want <= rows_of[geometry]) rc.rpb = want;
            }
            rc.G = (H + rc.rpb - 1) / rc.rpb;
            rc.H = H;
            rc.n_idx = live_index_bound(e, H);
            rc.iters_cap = cap;
            rc.spin_limit = e->spin_limit;
            rc.test_abort_epoch = e->test_abort_epoch;
            rc.test_late_wave0 = e->test_late_wave0;
            rc.dbg = nullptr;
#ifdef JSLP_DEBUG_RESIDENT
            static u64_t* dbg_buf = nullptr;
            if (!dbg_buf) HIPC(hipMalloc(&dbg_buf, sizeof(u64_t) * (512 * JSLP_F_MAXG * 2 + 16384)));
            HIPC(hipMemsetAsync(dbg_buf, 0, sizeof(u64_t) * (512 * JSLP_F_MAXG * 2 + 16384), s));
            rc.dbg = dbg_buf;
#endif
            HIPC(hipMemsetAsync(e->r_sync, 0, sizeof(unsigned) * 16, s));
            if (geometry == 6) {
                // the XCD-local build's rows carry their epoch tag INSIDE the data and every launch's tags restart at 1: what an earlier
                // launch left in the row slots would pass for this launch's rows (found by the Knapsack fixture's fifth relaxation: same
                // pivots, another RHS) -- the slots start from zero (tag 0 is never used)
                for (int i = 0; i < 2; i++) HIPC(hipMemsetAsync(e->r_rows[i], 0, (size_t)rc.G * e->ld * 16, s));
            }
            // Safety net.  The kernel commits every pivot's index-map swap as it goes but writes the matrix back only in
            // its epilogue, and a timed-out hand-off (workgroups not co-resident, a stalled GPU) skips that epilogue: keep a
            // copy of slot 0 (one pass over the matrix, ~15 us at 2001 x 2001 against a ~100 ms solve) to roll back to.
            SnapshotW bk{e->rb_A, e->rb_vibr, e->rb_vibc, e->rb_rbv, e->rb_cbv, e->n_idx, nullptr, nullptr};
            hipLaunchKernelGGL(k_res_backup, dim3(copy_grid(e, 1).x), dim3(256), 0, s, e->s, bk, e->r_backup_st, H, 1);
            hipEvent_t k0 = nullptr, k1 = nullptr;
            if (e->timing) { r = ensure_events(e, 2); if (r) return r; k0 = e->ev_pool[0]; k1 = e->ev_pool[1]; HIPC(hipEventRecord(k0, s)); }
            void* args[] = {&rc};
            // lane geometry: 1024 lanes x 2 columns, or 512 lanes x 4 columns (half the waves per workgroup barrier)
            // up to 8 rows per workgroup: 1024 lanes x 2 columns (or 512 x 4, measured slower); 9..16 rows (2048 < H <= 4096):
            // 512 lanes x 4 columns x 16 rows -- 2 waves per SIMD leave 256 VGPRs per lane for the 64 MB of tableau
            hipError_t le = hipErrorInvalidValue;
            const bool unr = e->n_unr > 0;  // unrestricted variables: the UNR build threads the per-column flags through
            // The LEAN build (all-gather protocol only, software-pipelined phase 2: jslp_resident_pipe.hip.h) takes every solve
            // without unrestricted variables; should its cycle-check history outgrow LDS it hands the solve over (status
            // ST_RUNNING / ST_PHASE1_DONE instead of ST_DONE) and the general build continues it in a second launch.
            bool lean = lean_on && lean_unr_ok;
#define JSLP_RES_LAUNCH_LEAN(T, C, R)                                                                                               \
    (unr ? (check_cycles ? hipLaunchCooperativeKernel((const void*)k_simplex_resident<T, C, R, true, true, false, true>, dim3(rc.G), dim3(T), args, 0, s)   \
                         : hipLaunchCooperativeKernel((const void*)k_simplex_resident<T, C, R, true, true, false, false>, dim3(rc.G), dim3(T), args, 0, s)) \
         : (check_cycles ? hipLaunchCooperativeKernel((const void*)k_simplex_resident<T, C, R, false, true, false, true>, dim3(rc.G), dim3(T), args, 0, s)  \
                         : hipLaunchCooperativeKernel((const void*)k_simplex_resident<T, C, R, false, true, false, false>, dim3(rc.G), dim3(T), args, 0, s)))
// (headline geometries: lean, else the general build)
#define JSLP_RES_LAUNCH(T, C, R)                                                                                                    \
    le = lean ? JSLP_RES_LAUNCH_LEAN(T, C, R)                                                                                       \
       : unr  ? hipLaunchCooperativeKernel((const void*)k_simplex_resident<T, C, R, true>, dim3(rc.G), dim3(T), args, 0, s)        \
              : hipLaunchCooperativeKernel((const void*)k_simplex_resident<T, C, R, false>, dim3(rc.G), dim3(T), args, 0, s)
// (tall / wide geometries: lean only -- a hand-over the lean build cannot continue goes to the streaming kernels)
#define JSLP_RES_LAUNCH_LEAN_ONLY(T, C, R)                                                                                          \
    do { if (lean) le = JSLP_RES_LAUNCH_LEAN(T, C, R); } while (0)
          resident_relaunch:
#if defined(JSLP_SPLIT_TU)  /* the product build: the instances live in jslp_tu_resident.hip (two parts, compiled side by side with this unit) */
            {
                static const int geo[7][3] = {{0, 0, 0}, {1024, 2, 8}, {512, 4, 8}, {512, 4, 16}, {512, 6, 12}, {512, 8, 8}, {512, 2, 32}};
                const int key[8] = {geo[geometry][0], geo[geometry][1], geo[geometry][2], unr ? 1 : 0, lean ? 1 : 0, e->n_opt > 0 ? 1 : 0, check_cycles ? 1 : 0, geometry == 6 ? 1 : 0};
                int r1 = jslpx_resident_launch_1(key, (unsigned)rc.G, &rc, s);
                if (r1 == -1) r1 = jslpx_resident_launch_2(key, (unsigned)rc.G, &rc, s);
                le = r1 == -1 ? hipErrorInvalidValue : (hipError_t)r1;  // (-1: no such instance -- what the switch below leaves `le` at)
            }
#elif defined(JSLP_DEV_HEADLINE_ONLY)  /* development builds: only the headline lean instances are compiled; never shipped */
            if (geometry != 1 || !lean || e->n_opt > 0) return fail(JSLP_ERR_UNSUPPORTED, "development build: headline lean geometry only");
            le = check_cycles ? hipLaunchCooperativeKernel((const void*)k_simplex_resident<1024, 2, 8, false, true, false, true>, dim3(rc.G), dim3(1024), args, 0, s)
                              : hipLaunchCooperativeKernel((const void*)k_simplex_resident<1024, 2, 8, false, true, false, false>, dim3(rc.G), dim3(1024), args, 0, s);
#elif defined(JSLP_DEV_TALL_ONLY)  /* development builds: only the tall lean instances <512, 4, 16> (register budget work; -DJSLP_DEV_TALL_ONLY=2: the OPT ones); never shipped */
            if (geometry != 3 || !lean || (e->n_opt > 0) != (JSLP_DEV_TALL_ONLY == 2) || unr) return fail(JSLP_ERR_UNSUPPORTED, "development build: tall lean geometry only");
            le = check_cycles ? hipLaunchCooperativeKernel((const void*)k_simplex_resident<512, 4, 16, false, true, JSLP_DEV_TALL_ONLY == 2, true>, dim3(rc.G), dim3(512), args, 0, s)
                              : hipLaunchCooperativeKernel((const void*)k_simplex_resident<512, 4, 16, false, true, JSLP_DEV_TALL_ONLY == 2, false>, dim3(rc.G), dim3(512), args, 0, s);
#elif defined(JSLP_DEV_XL_ONLY)  /* development builds: only the XCD-local instances are compiled (40 s instead of 2.5 min); never shipped */
            if (geometry != 6) return fail(JSLP_ERR_UNSUPPORTED, "development build: XCD-local geometry only");
            le = check_cycles ? hipLaunchCooperativeKernel((const void*)k_simplex_resident<512, 2, 32, false, true, false, true, true>, dim3(JSLP_XL_SPREAD * rc.G), dim3(512), args, 0, s)
                              : hipLaunchCooperativeKernel((const void*)k_simplex_resident<512, 2, 32, false, true, false, false, true>, dim3(JSLP_XL_SPREAD * rc.G), dim3(512), args, 0, s);
#else
            switch (geometry) {
                case 1:
                    if (e->n_opt > 0) {  // (resident_geometry admits optional objectives only here, and only for the lean build)
                        if (!lean) break;  // a hand-over the general build cannot take: the fused pipeline continues (below)
                        le = check_cycles ? hipLaunchCooperativeKernel((const void*)k_simplex_resident<1024, 2, 8, false, true, true, true>, dim3(rc.G), dim3(1024), args, 0, s)
                                          : hipLaunchCooperativeKernel((const void*)k_simplex_resident<1024, 2, 8, false, true, true, false>, dim3(rc.G), dim3(1024), args, 0, s);
                        break;
                    }
                    JSLP_RES_LAUNCH(1024, 2, 8);
                    break;
                case 2: JSLP_RES_LAUNCH(512, 4, 8); break;
                case 3:
                    if (e->n_opt > 0) {  // (lean build with the optional objective rows in registers; phase 1 came through the fused pipeline)
                        if (lean) le = check_cycles ? hipLaunchCooperativeKernel((const void*)k_simplex_resident<512, 4, 16, false, true, true, true>, dim3(rc.G), dim3(512), args, 0, s)
                                                    : hipLaunchCooperativeKernel((const void*)k_simplex_resident<512, 4, 16, false, true, true, false>, dim3(rc.G), dim3(512), args, 0, s);
                        break;
                    }
                    JSLP_RES_LAUNCH_LEAN_ONLY(512, 4, 16);
                    break;
                case 4: JSLP_RES_LAUNCH_LEAN_ONLY(512, 6, 12); break;
                case 5: JSLP_RES_LAUNCH_LEAN_ONLY(512, 8, 8); break;
#ifdef JSLP_WITH_XL
                case 6:  // XCD-local: every JSLP_XL_SPREAD-th block of the grid works (all of them on one XCD), the others return at once
                    if (!lean) break;  // (a hand-over the general build cannot take at this geometry: the streaming kernels continue)
                    le = check_cycles ? hipLaunchCooperativeKernel((const void*)k_simplex_resident<512, 2, 32, false, true, false, true, true>, dim3(JSLP_XL_SPREAD * rc.G), dim3(512), args, 0, s)
                                      : hipLaunchCooperativeKernel((const void*)k_simplex_resident<512, 2, 32, false, true, false, false, true>, dim3(JSLP_XL_SPREAD * rc.G), dim3(512), args, 0, s);
                    break;
#endif
            }
#endif
#undef JSLP_RES_LAUNCH
#undef JSLP_RES_LAUNCH_LEAN
#undef JSLP_RES_LAUNCH_LEAN_ONLY
            if (le == hipSuccess) e->resident_launches += 1;
            if (le == hipSuccess && lean) {
                // JSLP_INJECT_RESIDENT_ABORT_US=<n>: raise the host-abort word n microseconds into the launch -- the rollback path (kernel gives
                // up mid-solve, slot 0 restored from the safety-net copy, the solve re-run through the streaming kernels) exercised on the
                // library users load, whose kernels have no test hooks.  Read per launch (tests set it inside one process); once per solve.
                // (parsed strictly -- a stray or garbage value no longer aborts every resident solve -- and announced once)
                const char* inj = getenv("JSLP_INJECT_RESIDENT_ABORT_US");
                if (inj && *inj && !e->abort_injected) {
                    char* endp = nullptr;
                    const long us = strtol(inj, &endp, 10);
                    if (endp != inj && *endp == 0 && us >= 0 && us <= 10000000L) {
                        static std::atomic<int> told{0};
                        if (!told.exchange(1)) fprintf(stderr, "[jslp] JSLP_INJECT_RESIDENT_ABORT_US=%ld: every register-resident solve is aborted %ld us after its launch and re-run through the streaming kernels (fault injection)\n", us, us);
                        e->abort_injected = 1;
                        std::this_thread::sleep_for(std::chrono::microseconds(us));
                        __atomic_store_n(e->h_abort_ptr, 1u, __ATOMIC_RELEASE);
                    }
                }
            }
            if (le == hipSuccess && lean) {  // did the lean kernel finish the solve?
                unsigned* const h_retries = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(e->h_state) + sizeof(DevState) + 8);
                HIPC(hipMemcpyAsync(e->h_state, e->s.st, sizeof(DevState), hipMemcpyDeviceToHost, s));
                HIPC(hipMemcpyAsync(h_retries, e->r_sync + 5, sizeof(unsigned), hipMemcpyDeviceToHost, s));  // repeated looks of the checksummed row hand-over
                HIPC(hipStreamSynchronize(s));
                e->resident_fetch_retries += *h_retries;
                if (e->h_state->err == ERR_NONE && e->h_state->status != ST_DONE) {
                    e->resident_handovers += 1;
                    if (e->n_opt > 0 || geometry >= 3) {
                        // (XCD-local and tall / wide geometries: there is no general build of them)
                        // optional objectives: the general build does not take them -- the streaming kernels continue from the state the
                        // lean kernel left (tableau, maps and objective rows written back; status ST_RUNNING or ST_PHASE1_DONE)
                        handed_to_streaming = true;
                        if (e->timing) { float ms = 0; HIPC(hipEventRecord(k1, s)); HIPC(hipEventSynchronize(k1)); if (hipEventElapsedTime(&ms, k0, k1) == hipSuccess) e->upd_ms += ms; e->upd_launches += e->h_state->it1 + e->h_state->it2 - it_before; }
                        return JSLP_OK;
                    }
                    lean = false;
                    HIPC(hipMemsetAsync(e->r_gran, 0, sizeof(u64_t) * JSLP_R_SYNC_WORDS, s));  // the second launch's tags restart at 1
                    goto resident_relaunch;
                }
            }
            if (le == hipSuccess) {
                if (e->timing) HIPC(hipEventRecord(k1, s));
                HIPC(hipMemcpyAsync(e->h_state, e->s.st, sizeof(DevState), hipMemcpyDeviceToHost, s));
                HIPC(hipStreamSynchronize(s));
                if (e->timing) {
                    float ms = 0;
                    if (hipEventElapsedTime(&ms, k0, k1) == hipSuccess) e->upd_ms += ms;
                    e->upd_launches += e->h_state->it1 + e->h_state->it2 - it_before;  // unit = one pivot (16*H*W algorithmic bytes)
                }
                if (e->h_state->err == ERR_BARRIER) {
                    // roll slot 0 back to its state before the launch and solve through the streaming kernels instead
                    hipLaunchKernelGGL(k_res_backup, dim3(copy_grid(e, 1).x), dim3(256), 0, s, e->s, bk, e->r_backup_st, H, 0);
                    HIPC(hipGetLastError());
                    // ... and the host's copy of the state with it: the streaming loops below start from what h_state says (the tall /
                    // wide geometries come here with ST_PHASE1_DONE: without this the aborted kernel's ST_DONE + ERR_BARRIER ended the
                    // solve as a device error instead of finishing it through k_pivot_fused)
                    HIPC(hipMemcpyAsync(e->h_state, e->s.st, sizeof(DevState), hipMemcpyDeviceToHost, s));
                    HIPC(hipStreamSynchronize(s));
                    e->resident_fallbacks += 1;
                } else {
                resident_done = true;
                e->last_path = geometry == 6 ? "resident-xl" : "resident";
                if (e->counting) {  // a dense streaming update touches every cell: rows x columns per pivot
                    const long long piv = (long long)e->h_state->it1 + e->h_state->it2;
                    e->wc.gated_cells += piv * (long long)(H - 1) * e->W;
                    e->wc.gated_rows += piv * (long long)(H - 1);
                }
#ifdef JSLP_DEBUG_RESIDENT
                {   // phase timing + micro-costs of the debug build (tools/resident_phase_timing.py reads this file)
                    std::vector<u64_t> d(16384);
                    HIPC(hipMemcpy(d.data(), rc.dbg + (size_t)512 * rc.G * 2, sizeof(u64_t) * 16384, hipMemcpyDeviceToHost));
                    FILE* fb = fopen("gpurun_out/resident_r0.bin", "wb");
                    if (fb) { fwrite(d.data(), 8, 16384, fb); fclose(fb); }
                    HIPC(hipMemcpy(d.data(), rc.dbg, sizeof(u64_t) * 16384, hipMemcpyDeviceToHost));  // RT_STAMP: [8 epochs][MAXG][8 events]
                    fb = fopen("gpurun_out/resident_stamps.bin", "wb");
                    if (fb) { fwrite(d.data(), 8, 16384, fb); fclose(fb); }
                }
#endif
                }  // !ERR_BARRIER
            } else {
                (void)hipGetLastError();  // not co-resident on this device: use one launch per pivot instead
                e->resident_refusals += 1;
                if (getenv("JSLP_DEBUG_LAUNCH"))
                    fprintf(stderr, "[jslp] cooperative launch of k_simplex_resident refused (geometry %d, lean %d, unr %d, G %d): %s\n", geometry, (int)lean,
                            (int)(e->n_unr > 0), rc.G, hipGetErrorString(le));
            }
            return JSLP_OK;
        };
        // The headline geometries run the WHOLE solve register-resident (phase 1 included).  The tall / wide ones (H > 2048 or more
        // than 2048 columns) take only phase 2 there -- their phase-1 loop does not fit the 256 registers a lane has next to 64-72 MB
        // of tableau (it spills: 22 k pivots/s against 33.7 k through k_fused_p1 on 4001 x 2001) while the lean kernel's pipelined
        // phase 2 does (4 scratch loads per pivot): phase 1 goes through the fused pipeline first, which hands over with
        // ST_PHASE1_DONE.  r03_j: 4001 x 2001 113.4 k pivots/s against 35.7 k fused, 2501 x 2001 128.6 k against 48.5 k,
        // 2001 x 4001 75.5 k against 34.8 k.
        const bool resident_phase2_only = geometry >= 3 && geometry != 6;  // (6 = XCD-local: the whole solve, like the headline geometry)
        if (geometry != 0 && !resident_phase2_only) {
            int r = run_resident(0);
            if (r) return r;
        }
        if (!resident_done) {
        // ---- phase 1 through the fused pipeline (one launch per pivot; k_fused_p1), when the fused pipeline applies -----------
        bool p1_fused = false;
        const bool phase1_over = handed_to_streaming && e->h_state->status == ST_PHASE1_DONE;  // (h_state is current in that case only)
        if (phase1_over) p1_fused = true;
        if (!phase1_over && fused && fused_p1_on()) {
            int r = ensure_fused(e);
            if (r) return r;
            r = ensure_fused_oo(e);
            if (r) return r;
            const FusedCtx f = make_fused_ctx(e, c, H);
            void (*kp1)(FusedCtx, int) = e->ld <= JSLP_F_TW ? (e->n_unr > 0 ? k_fused_p1<1, true> : k_fused_p1<1, false>)
                                         : e->ld <= 2 * JSLP_F_TW ? (e->n_unr > 0 ? k_fused_p1<2, true> : k_fused_p1<2, false>)
                                         : e->ld <= 3 * JSLP_F_TW ? k_fused_p1<3, false> : k_fused_p1<4, false>;  // (fused_eligible: no unrestricted variables beyond two tiles)
            for (;;) {
                int launch = 0;
                hipLaunchKernelGGL(kp1, dim3(f.G), dim3(JSLP_F_THREADS), 0, s, f, launch);
                launch++;
                int chunk1 = 2;
                for (;;) {
                    for (int i = 0; i < chunk1; i++) { hipLaunchKernelGGL(kp1, dim3(f.G), dim3(JSLP_F_THREADS), 0, s, f, launch); launch++; }
                    HIPC(hipGetLastError());
                    HIPC(hipMemcpyAsync(e->h_state, f.fst[launch & 1], sizeof(DevState), hipMemcpyDeviceToHost, s));
                    HIPC(hipStreamSynchronize(s));
                    if (e->h_state->status != ST_RUNNING) break;
                    chunk1 = std::min(chunk1 * 2, 512);
                }
                hipLaunchKernelGGL(k_fused_finish, dim3(512), dim3(256), 0, s, f, launch - 1);  // state -> slot 0's, tableau -> buf[0]
                HIPC(hipGetLastError());
                if (e->h_state->status != ST_P1_SLOW) break;
                // the one pivot k_fused_p1 cannot decide alone (see there): k_select + k_update, then the pipeline again
                hipLaunchKernelGGL(k_p1_resume, dim3(1), dim3(1), 0, s, e->s.st, (int)ST_RUNNING);
                hipLaunchKernelGGL(k_select, dim3(1), dim3(JSLP_WG_THREADS), 0, s, c);
                hipLaunchKernelGGL(k_update, grid, dim3(JSLP_UPD_THREADS), 0, s, c);
                HIPC(hipGetLastError());
                HIPC(hipMemcpyAsync(e->h_state, e->s.st, sizeof(DevState), hipMemcpyDeviceToHost, s));
                HIPC(hipStreamSynchronize(s));
                e->p1_slow_pivots += 1;
                if (e->h_state->status != ST_RUNNING) break;
            }
            p1_fused = true;
            e->last_path = "fused";
        }
        // ---- phase 1 (and phase 2 when the fused pipeline does not apply): k_select + k_update per pivot ----
        int chunk = fused ? 1 : 8;
        long long done_prev = 0;
        for (; !p1_fused;) {
            if (e->timing && !fused) { int r = ensure_events(e, 2 * (size_t)chunk); if (r) return r; }
            for (int i = 0; i < chunk; i++) {
                hipLaunchKernelGGL(k_select, dim3(1), dim3(JSLP_WG_THREADS), 0, s, c);
                if (e->timing && !fused) HIPC(hipEventRecord(e->ev_pool[2 * i], s));
                hipLaunchKernelGGL(k_update, grid, dim3(JSLP_UPD_THREADS), 0, s, c);
                if (e->timing && !fused) HIPC(hipEventRecord(e->ev_pool[2 * i + 1], s));
            }
            HIPC(hipGetLastError());
            HIPC(hipMemcpyAsync(e->h_state, e->s.st, sizeof(DevState), hipMemcpyDeviceToHost, s));
            HIPC(hipStreamSynchronize(s));
            const DevState& st = *e->h_state;
            if (e->timing && !fused) {
                // launches after the solve ended are no-ops: only the first `real` of this chunk did a pivot
                const long long done = (long long)st.it1 + st.it2;
                const long long real = std::min<long long>(chunk, done - done_prev);
                for (long long i = 0; i < real; i++) {
                    float ms = 0;
                    if (hipEventElapsedTime(&ms, e->ev_pool[2 * i], e->ev_pool[2 * i + 1]) == hipSuccess) e->upd_ms += ms;
                }
                e->upd_launches += real;
                done_prev = done;
            }
            if (st.status != ST_RUNNING) break;
            chunk = std::min(chunk * 2, 256);
        }
        // ---- phase 2 of the tall / wide geometries: register-resident (see above) ------------------------------
        if (resident_phase2_only && e->h_state->status == ST_PHASE1_DONE && e->h_state->err == ERR_NONE) {
            int r = run_resident((long long)e->h_state->it1 + e->h_state->it2);
            if (r) return r;
        }
        // ---- phase 2: one fused launch per pivot ----------------------------------------------------------
        while (!resident_done && e->h_state->status == ST_PHASE1_DONE) {
            int r = ensure_fused(e);
            if (r) return r;
            r = ensure_fused_oo(e);
            if (r) return r;
            e->last_path = "fused";
            const FusedCtx f = make_fused_ctx(e, c, H);
            // column tiles per lane (ld <= 2048: one) x unrestricted variables present x optional objectives present
            const bool unr = e->n_unr > 0, opt = e->n_opt > 0;
            void (*kfused)(FusedCtx, int) =
                e->ld <= JSLP_F_TW ? (unr ? (opt ? k_pivot_fused<1, true, true> : k_pivot_fused<1, true, false>)
                                          : (opt ? k_pivot_fused<1, false, true> : k_pivot_fused<1, false, false>))
                : e->ld <= 2 * JSLP_F_TW ? (unr ? (opt ? k_pivot_fused<2, true, true> : k_pivot_fused<2, true, false>)
                                                : (opt ? k_pivot_fused<2, false, true> : k_pivot_fused<2, false, false>))
                : e->ld <= 3 * JSLP_F_TW ? k_pivot_fused<3, false, false> : k_pivot_fused<4, false, false>;  // (fused_eligible: plain tableaus only beyond two tiles)
            int launch = 0;
            hipLaunchKernelGGL(kfused, dim3(f.G), dim3(JSLP_F_THREADS), 0, s, f, launch);
            launch++;
            chunk = 16;
            long long it2_prev = e->h_state->it2;
            for (;;) {
                if (e->timing) { int r2 = ensure_events(e, 2 * (size_t)chunk); if (r2) return r2; }
                for (int i = 0; i < chunk; i++) {
                    if (e->timing) HIPC(hipEventRecord(e->ev_pool[2 * i], s));
                    hipLaunchKernelGGL(kfused, dim3(f.G), dim3(JSLP_F_THREADS), 0, s, f, launch);
                    if (e->timing) HIPC(hipEventRecord(e->ev_pool[2 * i + 1], s));
                    launch++;
                }
                HIPC(hipGetLastError());
                HIPC(hipMemcpyAsync(e->h_state, f.fst[launch & 1], sizeof(DevState), hipMemcpyDeviceToHost, s));
                HIPC(hipStreamSynchronize(s));
                const DevState& st = *e->h_state;
                if (e->timing) {
                    const long long real = std::min<long long>(chunk, (long long)st.it2 - it2_prev);
                    for (long long i = 0; i < real; i++) {
                        float ms = 0;
                        if (hipEventElapsedTime(&ms, e->ev_pool[2 * i], e->ev_pool[2 * i + 1]) == hipSuccess) e->upd_ms += ms;
                    }
                    e->upd_launches += real;
                    it2_prev = st.it2;
                }
                if (st.status != ST_RUNNING) break;
                chunk = std::min(chunk * 2, 512);
            }
            hipLaunchKernelGGL(k_fused_finish, dim3(512), dim3(256), 0, s, f, launch - 1);
            HIPC(hipGetLastError());
            if (e->h_state->status == ST_P1_SLOW) {
                // an entering column named by an optional objective with a ~0 main cost and a tiny pivot-row entry: that ONE pivot
                // through k_select + k_update (see k_pivot_fused), then the pipeline again from its first launch
                hipLaunchKernelGGL(k_p1_resume, dim3(1), dim3(1), 0, s, e->s.st, (int)ST_RUNNING);
                hipLaunchKernelGGL(k_select, dim3(1), dim3(JSLP_WG_THREADS), 0, s, c);
                hipLaunchKernelGGL(k_update, grid, dim3(JSLP_UPD_THREADS), 0, s, c);
                HIPC(hipGetLastError());
                HIPC(hipMemcpyAsync(e->h_state, e->s.st, sizeof(DevState), hipMemcpyDeviceToHost, s));
                HIPC(hipStreamSynchronize(s));
                e->p1_slow_pivots += 1;
                if (e->h_state->status != ST_RUNNING) break;  // that pivot ended the solve
                hipLaunchKernelGGL(k_p1_resume, dim3(1), dim3(1), 0, s, e->s.st, (int)ST_PHASE1_DONE);
                e->h_state->status = ST_PHASE1_DONE;
                continue;
            }
            HIPC(hipMemcpyAsync(e->h_state, e->s.st, sizeof(DevState), hipMemcpyDeviceToHost, s));
            if (e->counting) {  // the fused launches stream every cell (phase 1 went through k_select: counted on the device)
                e->wc.gated_cells += (long long)e->h_state->it2 * (long long)(H - 1) * e->W;
                e->wc.gated_rows += (long long)e->h_state->it2 * (long long)(H - 1);
            }
            HIPC(hipStreamSynchronize(s));
            break;
        }
        }  // !resident_done
        HIPC(hipEventRecord(e->ev_end, s));
        HIPC(hipStreamSynchronize(s));
    }
    float ms = 0;
    if (hipEventElapsedTime(&ms, e->ev_begin, e->ev_end) == hipSuccess) e->total_ms += ms;
    return state_error(*e->h_state);
}

extern "C" int jslp_engine_simplex(jslp_engine* e, int check_cycles, jslp_simplex_result* out) {
    if (!e || !out) return fail(JSLP_ERR_ARG, "simplex: null pointer");
    if (!e->uploaded) return fail(JSLP_ERR_STATE, "simplex before upload");
    HIPC(hipSetDevice(e->device));
    int rc = run_simplex(e, check_cycles);
    if (rc) return rc;
    account(e, *e->h_state, 0);
    return fill_result(e, *e->h_state, 0, e->evaluation, out, &e->evaluation);
}

extern "C" int jslp_engine_pivot(jslp_engine* e, int32_t row, int32_t col) {
    if (!e || !e->uploaded) return fail(JSLP_ERR_STATE, "pivot before upload");
    HIPC(hipSetDevice(e->device));
    hipStream_t s = e->stream;
    HIPC(hipMemcpyAsync(e->h_state, e->s.st, sizeof(DevState), hipMemcpyDeviceToHost, s));
    HIPC(hipStreamSynchronize(s));
    const int H = e->h_state->H;
    if (row < 0 || row >= H || col < 0 || col >= e->W) return fail(JSLP_ERR_ARG, "pivot: index out of range");
    e->slot0_synced = 0;  // k_prepare zeroes st.gen
    const Ctx c = host_ctx(e, 0);
    hipLaunchKernelGGL(k_prepare, dim3(1), dim3(JSLP_WG_THREADS), 0, s, c, (int)row, (int)col);
    hipLaunchKernelGGL(k_update, update_grid(e, H), dim3(JSLP_UPD_THREADS), 0, s, c);
    hipLaunchKernelGGL(k_end_pivot, dim3(1), dim3(1), 0, s, c);
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(s));
    return JSLP_OK;
}

static dim3 copy_grid(const jslp_engine* e, int slots) {
    const long long n2 = (long long)e->cap_rows * e->ld / 2;
    long long bx = std::max<long long>(8, std::min<long long>(256, 4096 / std::max(1, slots)));
    bx = std::min<long long>(bx, (n2 + 255) / 256);
    return dim3((unsigned)std::max<long long>(bx, 1), slots, 1);
}

// the saved root once more, column-major, for the node kernels' pivot-column reads (WgLds::snapT)
static void launch_snapshot_transpose(jslp_engine* e) {
    if (!e->snap_AT) return;
    hipLaunchKernelGGL(k_snapshot_transpose, dim3((e->W + 31) / 32, (e->cap_rows + 31) / 32), dim3(256), 0, e->stream, e->s.st, e->snap_A,
                       (int)e->ld, (int)e->W, e->snap_AT, e->snap_ldT);
}
static Snapshot root_snapshot(const jslp_engine* e) {
    return Snapshot{e->snap_A, e->snap_vibr, e->snap_vibc, e->snap_rbv, e->snap_cbv, e->n_idx, e->snap_oo, -1, 0, e->snap_rhs,
                    snapshot_transpose_on() ? e->snap_AT : nullptr, e->snap_ldT};
}

extern "C" int jslp_engine_save(jslp_engine* e) {
    if (!e || !e->uploaded) return fail(JSLP_ERR_STATE, "save before upload");
    HIPC(hipSetDevice(e->device));
    SnapshotW w{e->snap_A, e->snap_vibr, e->snap_vibc, e->snap_rbv, e->snap_cbv, e->n_idx, e->snap_oo, e->snap_rhs};
    e->slot0_synced = 0;  // new snapshot generation
    e->slots_synced = 0;
    hipLaunchKernelGGL(k_save, dim3(copy_grid(e, 1).x), dim3(256), 0, e->stream, e->s, w);
    launch_snapshot_transpose(e);
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(e->stream));
    e->has_save = 1;
    e->root_seq += 1;
    return JSLP_OK;
}

// restore slots [first_slot, first_slot + n) from the saved root (checkpoint < 0) or from a checkpoint
static int enqueue_restore(jslp_engine* e, int first_slot, int n, int checkpoint = -1) {
    if (first_slot == 0) {
        if (checkpoint < 0 && e->has_save) {
            e->slot0_synced = 1;
            e->slots_synced = std::max(e->slots_synced, n);
        } else {
            e->slot0_synced = 0;
            if (checkpoint >= 0) e->slots_synced = 0;  // these slots now hold a checkpoint, not the snapshot
        }
    }
    if (checkpoint < 0) {
        if (!e->has_save) return JSLP_OK;  // backup.ts:54-56
        Snapshot sn = root_snapshot(e);
        hipLaunchKernelGGL(k_restore, copy_grid(e, n), dim3(256), 0, e->stream, e->s, sn, first_slot);
    } else {
        const jslp_engine::Ckpt& c = e->ckpts[checkpoint];
        Snapshot sn{c.A, c.vibr, c.vibc, c.rbv, c.cbv, e->n_idx, nullptr, c.H, c.last_element_index, c.rhs};
        hipLaunchKernelGGL(k_restore, copy_grid(e, n), dim3(256), 0, e->stream, e->s, sn, first_slot);
    }
    hipLaunchKernelGGL(k_restore_commit, dim3(n), dim3(1), 0, e->stream, e->s, first_slot, checkpoint < 0 ? 1 : 0);
    HIPC(hipGetLastError());
    return JSLP_OK;
}

static int checkpoint_check(const jslp_engine* e, int32_t id, const char* who) {
    if (!e || !e->uploaded) return fail(JSLP_ERR_STATE, "%s%s", who, ": before upload");
    if (id < 0 || id >= (int32_t)e->ckpts.size() || !e->ckpts[id].live) return fail(JSLP_ERR_ARG, "%s%s", who, ": no such checkpoint");
    return JSLP_OK;
}

extern "C" int jslp_engine_checkpoint_create(jslp_engine* e, int32_t* id_out) {
    if (!e || !id_out) return fail(JSLP_ERR_ARG, "checkpoint_create: null pointer");
    if (!e->uploaded) return fail(JSLP_ERR_STATE, "checkpoint_create before upload");
    HIPC(hipSetDevice(e->device));
    // height / lastElementIndex of the live tableau
    HIPC(hipMemcpyAsync(e->h_state, e->s.st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
    HIPC(hipStreamSynchronize(e->stream));
    jslp_engine::Ckpt c;
    for (int pass = 0; pass < 2; pass++) {  // every buffer is sized for the row capacity, so freed ones fit any later checkpoint
        Carver cv{pass ? c.mem : nullptr, 0};
        c.A = cv.take<double>((size_t)e->cap_rows * e->ld);
        c.rhs = cv.take<double>((size_t)e->cap_rows);
        c.vibr = cv.take<int32_t>((size_t)e->cap_rows);
        c.vibc = cv.take<int32_t>((size_t)e->W);
        c.rbv = cv.take<int32_t>((size_t)e->n_idx);
        c.cbv = cv.take<int32_t>((size_t)e->n_idx);
        if (!pass) {
            if (!e->ck_free.empty()) {
                c.mem = e->ck_free.back();
                e->ck_free.pop_back();
            } else if (hipMalloc(&c.mem, cv.off + 256) != hipSuccess) {
                (void)hipGetLastError();
                return fail(JSLP_ERR_NOMEM, "checkpoint_create: out of device memory");
            }
        }
    }
    c.H = e->h_state->H;
    c.last_element_index = e->h_state->last_element_index;
    c.evaluation = e->evaluation;
    c.live = 1;
    SnapshotW w{c.A, c.vibr, c.vibc, c.rbv, c.cbv, e->n_idx, nullptr, c.rhs};
    hipLaunchKernelGGL(k_checkpoint, dim3(copy_grid(e, 1).x), dim3(256), 0, e->stream, e->s, w, (int)c.H);
    HIPC(hipGetLastError());
    int32_t id = -1;
    for (size_t i = 0; i < e->ckpts.size(); i++)
        if (!e->ckpts[i].live) { id = (int32_t)i; break; }
    if (id < 0) { id = (int32_t)e->ckpts.size(); e->ckpts.push_back(c); } else e->ckpts[id] = c;
    *id_out = id;
    return JSLP_OK;  // stream-ordered: no synchronisation needed before the next engine call
}

extern "C" int jslp_engine_checkpoint_restore(jslp_engine* e, int32_t id) {
    int rc = checkpoint_check(e, id, "checkpoint_restore");
    if (rc) return rc;
    HIPC(hipSetDevice(e->device));
    rc = enqueue_restore(e, 0, 1, id);
    if (rc) return rc;
    HIPC(hipStreamSynchronize(e->stream));
    e->evaluation = e->ckpts[id].evaluation;  // incremental-branch-and-cut.ts:105
    return JSLP_OK;
}

extern "C" int jslp_engine_checkpoint_release(jslp_engine* e, int32_t id) {
    int rc = checkpoint_check(e, id, "checkpoint_release");
    if (rc) return rc;
    HIPC(hipSetDevice(e->device));
    HIPC(hipStreamSynchronize(e->stream));  // a restore from it may still be in flight
    e->ck_free.push_back(e->ckpts[id].mem);
    e->ckpts[id] = jslp_engine::Ckpt();
    return JSLP_OK;
}

extern "C" int jslp_engine_restore(jslp_engine* e) {
    if (!e || !e->uploaded) return fail(JSLP_ERR_STATE, "restore before upload");
    HIPC(hipSetDevice(e->device));
    int rc = enqueue_restore(e, 0, 1);
    if (rc) return rc;
    HIPC(hipStreamSynchronize(e->stream));
    return JSLP_OK;
}

// stage the cut lists of n_nodes nodes on the device: packed into one pinned buffer, one async copy, no sync
// (the pinned buffer is reused only after the call's final stream synchronisation)
static int upload_cuts(jslp_engine* e, int32_t n_nodes, const int32_t* offs, const int8_t* type, const int32_t* var,
                       const double* value, bool to_device = true) {
    if (offs[0] != 0) return fail(JSLP_ERR_ARG, "cuts: cut_offsets[0] must be 0");
    for (int32_t i = 0; i < n_nodes; i++)
        if (offs[i + 1] < offs[i]) return fail(JSLP_ERR_ARG, "cuts: cut_offsets must not decrease");
    const size_t C = (size_t)offs[n_nodes], N1 = (size_t)n_nodes + 1;
    if (C > 0 && (!type || !var || !value)) return fail(JSLP_ERR_ARG, "cuts: null pointer");
    // [values | offsets | variables | types | hand-out order of a batch | its queue counter (0)]
    const size_t off_offs = 8 * C, off_var = off_offs + 4 * N1, off_type = off_var + 4 * C, off_order = (off_type + C + 3) & ~(size_t)3,
                 off_queue = off_order + 4 * (size_t)n_nodes, total = off_queue + 4;
    if (total > e->cuts_bytes) {  // grow: allocate the new pair first, swap only when both exist
        const size_t bytes = std::max<size_t>(2 * total, 4096);
        char* nd = nullptr; char* nh = nullptr;
        if (hipMalloc(&nd, bytes) != hipSuccess) { (void)hipGetLastError(); return fail(JSLP_ERR_NOMEM, "cuts: out of device memory"); }
        if (hipHostMalloc(&nh, bytes) != hipSuccess) { (void)hipGetLastError(); hipFree(nd); return fail(JSLP_ERR_NOMEM, "cuts: out of pinned memory"); }
        HIPC(hipStreamSynchronize(e->stream));  // nothing in flight may still read the old pair
        hipFree(e->d_cuts);
        if (e->h_cuts) hipHostFree(e->h_cuts);
        e->d_cuts = nd; e->h_cuts = nh; e->cuts_bytes = bytes;
    }
    if (C) memcpy(e->h_cuts, value, 8 * C);
    memcpy(e->h_cuts + off_offs, offs, 4 * N1);
    if (C) { memcpy(e->h_cuts + off_var, var, 4 * C); memcpy(e->h_cuts + off_type, type, C); }
    if (node_queue() == 2) {  // most cuts first (a counting sort, stable): k_node_queue hands the nodes out in this order
        int32_t* order = reinterpret_cast<int32_t*>(e->h_cuts + off_order);
        int most = 0;
        for (int32_t i = 0; i < n_nodes; i++) most = std::max(most, (int)(offs[i + 1] - offs[i]));
        std::vector<int32_t>& start = e->order_scratch;
        start.assign((size_t)most + 2, 0);
        for (int32_t i = 0; i < n_nodes; i++) start[(size_t)(most - (offs[i + 1] - offs[i])) + 1]++;
        for (int c = 0; c <= most; c++) start[(size_t)c + 1] += start[c];
        for (int32_t i = 0; i < n_nodes; i++) order[start[(size_t)(most - (offs[i + 1] - offs[i]))]++] = i;
    }
    *reinterpret_cast<int32_t*>(e->h_cuts + off_queue) = 0;
    // to_device == false: the one-launch node kernel reads the (tiny) cut list straight from the pinned buffer
    char* base = e->d_cuts;
    if (to_device) {
        HIPC(hipMemcpyAsync(e->d_cuts, e->h_cuts, total, hipMemcpyHostToDevice, e->stream));
    } else {
        void* dev = nullptr;
        HIPC(hipHostGetDevicePointer(&dev, e->h_cuts, 0));
        base = static_cast<char*>(dev);
    }
    e->d_cut_val = reinterpret_cast<double*>(base);
    e->d_cut_offs = reinterpret_cast<int32_t*>(base + off_offs);
    e->d_cut_var = reinterpret_cast<int32_t*>(base + off_var);
    e->d_cut_type = reinterpret_cast<int8_t*>(base + off_type);
    e->d_cut_order = reinterpret_cast<int32_t*>(base + off_order);
    e->d_queue = reinterpret_cast<int*>(base + off_queue);
    return JSLP_OK;
}

extern "C" int jslp_engine_add_cuts(jslp_engine* e, int32_t n, const int8_t* type, const int32_t* var_index,
                                    const double* value) {
    if (!e || !e->uploaded) return fail(JSLP_ERR_STATE, "add_cuts before upload");
    if (n < 0) return fail(JSLP_ERR_ARG, "add_cuts: negative count");
    HIPC(hipSetDevice(e->device));
    const int32_t offs[2] = {0, n};
    int rc = upload_cuts(e, 1, offs, type, var_index, value);
    if (rc) return rc;
    if (e->counting) e->wc.cut_rows += n;
    Cuts cu{e->d_cut_offs, e->d_cut_type, e->d_cut_var, e->d_cut_val};
    hipLaunchKernelGGL(k_add_cuts, dim3(1), dim3(256), 0, e->stream, e->s, cu, 0, 0, (int)e->cap_rows);
    HIPC(hipGetLastError());
    HIPC(hipMemcpyAsync(e->h_state, e->s.st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
    HIPC(hipStreamSynchronize(e->stream));
    return state_error(*e->h_state);
}

// read-back buffers for `nodes` nodes laid out [states | rhs | rows] so that one copy brings a group back
static size_t out_bytes(const jslp_engine* e, size_t nodes) {
    return nodes * (sizeof(DevState) + (((size_t)e->cap_rows + 3) & ~(size_t)3) * 12) + 192;
}
// rows per node in the engine's own read-back buffers: a multiple of 4, so that every node's slice starts 16-byte aligned and
// the kernels write it with 8/16-byte stores per lane (4-byte stores into pinned host memory cost the Monster_II batch a
// quarter of its time); the device pool lays its shared buffer out itself, with the plain row capacity
static size_t out_stride_of(const jslp_engine* e) {
    return e->ext_states ? (size_t)e->cap_rows : (((size_t)e->cap_rows + 3) & ~(size_t)3);
}
static void out_layout(jslp_engine* e, size_t nodes) {
    const size_t stride = out_stride_of(e);
    const size_t o_rhs = (nodes * sizeof(DevState) + 63) & ~(size_t)63, o_rows = (o_rhs + nodes * stride * 8 + 63) & ~(size_t)63;
    e->d_states = reinterpret_cast<DevState*>(e->d_out);
    e->d_rhs = reinterpret_cast<double*>(e->d_out + o_rhs);
    e->d_rows = reinterpret_cast<int32_t*>(e->d_out + o_rows);
    e->h_states = reinterpret_cast<DevState*>(e->h_out);
    e->h_rhs = reinterpret_cast<double*>(e->h_out + o_rhs);
    e->h_rows = reinterpret_cast<int32_t*>(e->h_out + o_rows);
    if (e->ext_states) {  // device pool: this call's outcomes go straight into the pool's pinned buffer
        e->h_states = e->ext_states; e->h_rhs = e->ext_rhs; e->h_rows = e->ext_rows;
    }
}
static int ensure_out(jslp_engine* e, size_t nodes) {
    if (out_bytes(e, nodes) > e->out_bytes_cap) {
        hipFree(e->d_out);
        if (e->h_out) hipHostFree(e->h_out);
        e->d_out = e->h_out = nullptr;
        e->out_bytes_cap = 0;
        const size_t bytes = out_bytes(e, std::max<size_t>(nodes, 16));
        HIPC(hipMalloc(&e->d_out, bytes));
        HIPC(hipHostMalloc(&e->h_out, bytes));
        e->out_bytes_cap = bytes;
    }
    out_layout(e, nodes);
    return JSLP_OK;
}

extern "C" int jslp_engine_read_rhs(jslp_engine* e, double* rhs, int32_t* var_index_by_row) {
    if (!e || !e->uploaded) return fail(JSLP_ERR_STATE, "read_rhs before upload");
    HIPC(hipSetDevice(e->device));
    int rc = ensure_out(e, 1);
    if (rc) return rc;
    hipStream_t s = e->stream;
    hipLaunchKernelGGL(k_gather, dim3(1), dim3(256), 0, s, e->s, 0, e->d_rhs, e->d_rows, e->d_states, (int)e->cap_rows, 0);
    HIPC(hipGetLastError());
    HIPC(hipMemcpyAsync(e->h_out, e->d_out, out_bytes(e, 1), hipMemcpyDeviceToHost, s));
    HIPC(hipStreamSynchronize(s));
    const int H = e->h_states[0].H;
    if (rhs) memcpy(rhs, e->h_rhs, sizeof(double) * H);
    if (var_index_by_row) memcpy(var_index_by_row, e->h_rows, sizeof(int32_t) * H);
    return JSLP_OK;
}

extern "C" int jslp_engine_set_integer_variables(jslp_engine* e, const int32_t* var_indexes, int32_t n) {
    if (!e || !e->uploaded) return fail(JSLP_ERR_STATE, "set_integer_variables before upload");
    if (n < 0 || (n > 0 && !var_indexes)) return fail(JSLP_ERR_ARG, "set_integer_variables: bad arguments");
    HIPC(hipSetDevice(e->device));
    std::vector<uint8_t> flags(e->n_idx, 0);
    for (int32_t i = 0; i < n; i++) {
        if (var_indexes[i] < 0 || var_indexes[i] >= e->n_idx) return fail(JSLP_ERR_ARG, "set_integer_variables: index out of range");
        flags[var_indexes[i]] = 1;
    }
    HIPC(hipMemcpyAsync(e->d_isint, flags.data(), e->n_idx, hipMemcpyHostToDevice, e->stream));
    HIPC(hipStreamSynchronize(e->stream));
    return JSLP_OK;
}

// Tableau.applyMIRCuts() (cutting-strategies.ts:199-212)
extern "C" int jslp_engine_apply_mir_cuts(jslp_engine* e, int32_t* n_added) {
    if (!e || !e->uploaded) return fail(JSLP_ERR_STATE, "apply_mir_cuts before upload");
    HIPC(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_mir_cuts, dim3(1), dim3(256), 0, e->stream, e->s, e->d_isint, 10, (int)e->cap_rows);
    HIPC(hipGetLastError());
    HIPC(hipMemcpyAsync(e->h_state, e->s.st, sizeof(DevState), hipMemcpyDeviceToHost, e->stream));
    HIPC(hipStreamSynchronize(e->stream));
    int rc = state_error(*e->h_state);
    if (rc) return rc;
    if (n_added) *n_added = e->h_state->mir_added;
    return JSLP_OK;
}

extern "C" int jslp_engine_mir_round(jslp_engine* e, int check_cycles, int32_t* n_added, jslp_simplex_result* out,
                                     double* rhs, int32_t* var_index_by_row) {
    int rc = jslp_engine_apply_mir_cuts(e, n_added);
    if (rc) return rc;
    rc = jslp_engine_simplex(e, check_cycles, out);
    if (rc) return rc;
    return jslp_engine_read_rhs(e, rhs, var_index_by_row);
}

// ---- fp32 twin -----------------------------------------------------------------------------------------------------
static int ensure_f32(jslp_engine* e) {
    if (e->arena32) return JSLP_OK;
    f32::Slots& s = e->s32;
    s.A_stride = (long long)e->cap_rows * e->ld;
    s.vibr_stride = e->cap_rows; s.vibc_stride = e->W; s.idx_stride = e->n_idx;
    s.prow_stride = e->ld; s.pcol_stride = e->cap_rows;
    s.hist_cap = (int32_t)HIST_CAP_SLOT;
    s.ld = e->ld; s.W = e->W; s.batch = e->batch; s.use_partial = e->use_partial;
    s.oo = nullptr; s.oo_stride = 0; s.n_opt = 0;
    s.rhs = nullptr;
    s.trace = nullptr; s.trace_cap = 0;
    for (int pass = 0; pass < 2; pass++) {
        Carver cv{pass ? e->arena32 : nullptr, 0};
        s.A = cv.take<float>((size_t)s.A_stride);
        s.vibr = cv.take<int32_t>((size_t)s.vibr_stride);
        s.vibc = cv.take<int32_t>((size_t)s.vibc_stride);
        s.rbv = cv.take<int32_t>((size_t)s.idx_stride);
        s.cbv = cv.take<int32_t>((size_t)s.idx_stride);
        s.prow = cv.take<float>((size_t)s.prow_stride);
        s.pcol = cv.take<float>((size_t)s.pcol_stride);
        s.dirty = cv.take<uint8_t>((size_t)s.pcol_stride);
        s.st = cv.take<DevState>(1);
        s.hist = cv.take<int2>((size_t)s.hist_cap);
        if (!pass) HIPC(hipMalloc(&e->arena32, cv.off + 256));
    }
    HIPC(hipMemsetAsync(e->arena32, 0, sizeof(float) * (size_t)s.A_stride, e->stream));  // padding columns / rows stay 0
    return JSLP_OK;
}

extern "C" int jslp_engine_simplex_f32(jslp_engine* e, double precision, int check_cycles, jslp_simplex_result* out,
                                       double* rhs, int32_t* var_index_by_row, double* device_ms) {
    if (!e || !out) return fail(JSLP_ERR_ARG, "simplex_f32: null pointer");
    if (!e->uploaded) return fail(JSLP_ERR_STATE, "simplex_f32 before upload");
    if (e->n_opt > 0) return fail(JSLP_ERR_UNSUPPORTED, "simplex_f32: optional objectives are not part of the fp32 experiment");
    if (!(precision > 0)) return fail(JSLP_ERR_ARG, "simplex_f32: precision must be positive");
    HIPC(hipSetDevice(e->device));
    int rc = ensure_f32(e);
    if (rc) return rc;
    rc = ensure_out(e, 1);
    if (rc) return rc;
    hipStream_t s = e->stream;
    f32::Slots& d = e->s32;
    d.unr = e->d_unr;
    d.has_unr = e->n_unr > 0 ? 1 : 0;
    d.precision = (float)precision;
    hipLaunchKernelGGL(k32_convert, dim3(copy_grid(e, 1).x), dim3(256), 0, s, e->s, d, 0);
    hipLaunchKernelGGL(f32::k_begin, dim3(1), dim3(1), 0, s, d, 0, iters_cap(e));
    HIPC(hipGetLastError());
    HIPC(hipMemcpyAsync(e->h_state, d.st, sizeof(DevState), hipMemcpyDeviceToHost, s));
    HIPC(hipStreamSynchronize(s));
    const int H = e->h_state->H;
    f32::Ctx c{};  // (cnt = nullptr: the fp32 experiment is not counted)
    c.A = d.A; c.vibr = d.vibr; c.vibc = d.vibc; c.rbv = d.rbv; c.cbv = d.cbv; c.unr = d.unr;
    c.prow = d.prow; c.pcol = d.pcol; c.dirty = d.dirty; c.rhs = nullptr; c.oo = nullptr; c.n_opt = 0; c.st = d.st; c.hist = d.hist;
    c.hist_cap = d.hist_cap; c.trace = nullptr; c.trace_cap = 0; c.ld = e->ld; c.W = e->W; c.check_cycles = check_cycles;
    c.batch = e->batch; c.use_partial = e->use_partial; c.precision = (float)precision; c.stop_at_phase2 = 0;
    c.has_unr = d.has_unr;
    const dim3 grid = update_grid(e, H);
    HIPC(hipEventRecord(e->ev_begin, s));
    for (int chunk = 8;; chunk = std::min(chunk * 2, 256)) {
        for (int i = 0; i < chunk; i++) {
            hipLaunchKernelGGL(f32::k_select, dim3(1), dim3(JSLP_WG_THREADS), 0, s, c);
            hipLaunchKernelGGL(f32::k_update, grid, dim3(JSLP_UPD_THREADS), 0, s, c);
        }
        HIPC(hipGetLastError());
        HIPC(hipMemcpyAsync(e->h_state, d.st, sizeof(DevState), hipMemcpyDeviceToHost, s));
        HIPC(hipStreamSynchronize(s));
        if (e->h_state->status != ST_RUNNING) break;
    }
    HIPC(hipEventRecord(e->ev_end, s));
    hipLaunchKernelGGL(k32_gather, dim3(1), dim3(256), 0, s, d, e->d_rhs, e->d_rows, e->d_states);
    HIPC(hipGetLastError());
    HIPC(hipMemcpyAsync(e->h_out, e->d_out, out_bytes(e, 1), hipMemcpyDeviceToHost, s));
    HIPC(hipStreamSynchronize(s));
    const DevState st = e->h_states[0];
    rc = state_error(st);
    if (rc) return rc;
    if (device_ms) {
        float ms = 0;
        *device_ms = hipEventElapsedTime(&ms, e->ev_begin, e->ev_end) == hipSuccess ? ms : -1.0;
    }
    memset(out, 0, sizeof *out);
    out->feasible = st.feasible;
    out->bounded = st.bounded;
    out->optimal = st.optimal;
    out->unbounded_var_index = st.bounded ? -1 : st.unbounded_var;
    out->pivots_phase1 = st.it1;
    out->pivots_phase2 = st.entered_phase2 ? st.it2 : -1;
    out->cycle_phase = st.cycle_phase;
    out->height = st.H;
    out->obj_cell = st.obj_cell;
    out->evaluation = e->evaluation;
    if (st.optimal) {
        const double rcoef = js_round(1.0 / precision);
        out->evaluation = js_round((2.220446049250313e-16 + st.obj_cell) * rcoef) / rcoef;
    } else if (!st.bounded) {
        out->evaluation = -INFINITY;
    }
    if (rhs) memcpy(rhs, e->h_rhs, sizeof(double) * st.H);
    if (var_index_by_row) memcpy(var_index_by_row, e->h_rows, sizeof(int32_t) * st.H);
    return JSLP_OK;
}

// Shared body of relax_batch / relax_batch_pinned.  Per-node-workgroup path: all groups are enqueued back to back,
// their outcomes accumulate in ONE device buffer laid out for all n_nodes, one copy and one synchronisation end the
// call.  `pinned` != 0: the caller reads the pinned buffer itself (no second host copy).
static int relax_batch_impl(jslp_engine* e, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                            const int32_t* var_index, const double* value, int check_cycles, jslp_simplex_result* out,
                            double* rhs, int32_t* var_index_by_row, int32_t out_stride, int pinned, int want_rhs,
                            int want_rows, int checkpoint = -1, int compact = 0) {
    const auto t_enter = std::chrono::steady_clock::now();  // (JSLP_DEBUG_STALL)
    const bool dev_out = e && e->dev_states != nullptr;  // outcomes stay on the device (jslp_engine_relax_batch_device)
    if (e && !dev_out) e->dev_prev_valid = 0;
    if (!e || n_nodes < 0 || !cut_offsets || (!out && !dev_out)) return fail(JSLP_ERR_ARG, "relax_batch: null pointer");
    if (!e->uploaded) return fail(JSLP_ERR_STATE, "relax before upload");
    if (compact && (e->n_watch <= 0 || e->n_watch > e->cap_rows))
        return fail(JSLP_ERR_ARG, "relax_watched: after set_watched_variables (at most row_capacity of them)");
    // gather mode: >= 0 = the whole RHS column / row map with this row stride; < 0 = the watched variables only
    const int g_stride = compact ? -e->n_watch : (dev_out ? (int)e->dev_stride : (int)out_stride_of(e));
    const size_t row_stride = compact ? (size_t)e->n_watch : (dev_out ? (size_t)e->dev_stride : out_stride_of(e));  // entries per node in the read-back buffers
    if (checkpoint >= 0) {
        int rc0 = checkpoint_check(e, checkpoint, "relax_from");
        if (rc0) return rc0;
    }
    if (!pinned && (rhs || var_index_by_row) && out_stride < e->cap_rows)
        return fail(JSLP_ERR_ARG, "relax_batch: out_stride < row capacity");
    if (n_nodes == 0) return JSLP_OK;
    if (n_nodes > 1 && !e->has_save && checkpoint < 0)
        return fail(JSLP_ERR_STATE, "relax_batch: several nodes need a saved root (save() first)");
    HIPC(hipSetDevice(e->device));
    hipStream_t s = e->stream;
    const int cap = iters_cap(e);
    const long long cells = (long long)e->cap_rows * e->ld;
    int rc;
    // ---- ONE child of the saved root, slot 0 already in sync with the snapshot: one launch, one synchronisation ----------
    if (n_nodes == 1 && checkpoint < 0 && e->has_save && e->slot0_synced && !e->timing && e->force_path <= 1 &&
        e->one_launch_nodes && cells <= wg_cells_child() && !e->ext_states && !dev_out) {
        rc = upload_cuts(e, 1, cut_offsets, type, var_index, value, false);
        if (rc) return rc;
        rc = ensure_out(e, 1);
        if (rc) return rc;
        void* out_dev = nullptr;
        HIPC(hipHostGetDevicePointer(&out_dev, e->h_out, 0));
        char* ob = static_cast<char*>(out_dev);
        DevState* o_state = reinterpret_cast<DevState*>(ob);
        double* o_rhs = want_rhs ? reinterpret_cast<double*>(ob + ((char*)e->h_rhs - e->h_out)) : nullptr;
        int32_t* o_rows = want_rows ? reinterpret_cast<int32_t*>(ob + ((char*)e->h_rows - e->h_out)) : nullptr;
        Cuts cu{e->d_cut_offs, e->d_cut_type, e->d_cut_var, e->d_cut_val};
        Snapshot sn = root_snapshot(e);
        e->last_path = "workgroup";
        unsigned* h_flag = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(e->h_state) + sizeof(DevState));
        void* flag_dev = nullptr;
        HIPC(hipHostGetDevicePointer(&flag_dev, h_flag, 0));
        unsigned* d_flag = static_cast<unsigned*>(flag_dev);
        const unsigned seq = ++e->done_seq ? e->done_seq : ++e->done_seq;  // never 0 (the flag's initial value)
        if (const size_t lds = wglds_smem(e); lds && e->s.n_opt > 0)
            hipLaunchKernelGGL((k_node_lds<JSLP_WG_THREADS, true>), dim3(1), dim3(JSLP_WG_THREADS), lds, s, e->s, sn, cu, 0, check_cycles,
                               cap, (int)e->cap_rows, o_rhs, o_rows, o_state, compact ? g_stride : 0, 0, d_flag, seq);
        else if (lds && node_cow_single())  // copy-on-write start, slot 0 made whole again behind the completion flag (jslp_wglds.hip.h)
            hipLaunchKernelGGL((k_node_lds<JSLP_WG_THREADS, false, true>), dim3(1), dim3(JSLP_WG_THREADS), lds, s, e->s, sn, cu, 0, check_cycles,
                               cap, (int)e->cap_rows, o_rhs, o_rows, o_state, compact ? g_stride : 0, 0, d_flag, seq);
        else if (lds)
            hipLaunchKernelGGL((k_node_lds<JSLP_WG_THREADS>), dim3(1), dim3(JSLP_WG_THREADS), lds, s, e->s, sn, cu, 0, check_cycles,
                               cap, (int)e->cap_rows, o_rhs, o_rows, o_state, compact ? g_stride : 0, 0, d_flag, seq);
        else
            hipLaunchKernelGGL((k_node_wg<JSLP_WG_THREADS, 4096>), dim3(1), dim3(JSLP_WG_THREADS), 0, s, e->s, sn, cu, 0, check_cycles,
                               cap, (int)e->cap_rows, o_rhs, o_rows, o_state, compact ? g_stride : 0, 0, d_flag, seq);
        HIPC(hipGetLastError());
        // the kernel's last act is a system-scope release store of `seq` into pinned memory: poll it (a stream
        // synchronisation costs tens of microseconds of wake-up latency per node of a sequential tree walk)
        {
            unsigned spins = 0;
            const auto t_begin = std::chrono::steady_clock::now();
            bool arrived = false;
            while (!(arrived = (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) == seq))) {
                if ((++spins & 0x3fffu) == 0 &&
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() > 5.0) break;
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
            }
            if (!arrived) HIPC(hipStreamSynchronize(s));  // a fault or a hang: let the runtime report it
        }
        const DevState st = e->h_states[0];
        rc = state_error(st);
        if (rc) { e->slot0_synced = 0; return rc; }
        account(e, st, 1);
        if (e->counting) e->wc.cut_rows += cut_offsets[1];
        double ev;
        rc = fill_result(e, st, 0, e->evaluation, &out[0], &ev);
        if (rc) return rc;
        e->evaluation = ev;
        if (!pinned) {
            const size_t n_out = compact ? (size_t)e->n_watch : (size_t)st.H;
            if (rhs) memcpy(rhs, e->h_rhs, sizeof(double) * n_out);
            if (var_index_by_row) memcpy(var_index_by_row, e->h_rows, sizeof(int32_t) * n_out);
        }
        return JSLP_OK;
    }
    rc = upload_cuts(e, n_nodes, cut_offsets, type, var_index, value);
    if (rc) return rc;
    Cuts cu{e->d_cut_offs, e->d_cut_type, e->d_cut_var, e->d_cut_val};
    // branch-and-bound children (a saved root exists) need a handful of repair pivots each: one workgroup, one launch,
    // no host round trip.  A first solve / plain LP goes through the chip-wide path unless the tableau is tiny.
    const bool wg = e->force_path == 1 ||
                    (e->force_path == 0 && (((e->has_save || checkpoint >= 0) && cells <= (n_nodes > 1 ? WG_CELLS_BATCH : wg_cells_child())) || use_wg_single(e)));
    // a node that does not reach an optimum keeps the evaluation it started with: the checkpoint's (restoreCheckpoint,
    // incremental-branch-and-cut.ts:105) or the live one (restore() leaves it alone)
    const double prev_eval = checkpoint >= 0 ? e->ckpts[checkpoint].evaluation : e->evaluation;
    // group size: bounded by memory (<= 16 GiB of the 288 GB for tableau copies) and by what keeps every CU busy with several nodes
    int group = 1;
    if (wg) {
        const long long max_slots = std::max<long long>(1, (16LL << 30) / (cells * 8));
        group = (int)std::min<long long>(std::min<long long>(n_nodes, group_max()), max_slots);
        if (const size_t lds = wglds_smem(e); lds && node_queue() && wg_batch_threads() == 512 && n_nodes > 1) {
            // the queue kernel wants exactly as many slots as the chip keeps workgroups resident
            if (e->queue_wgs == 0 || e->queue_wgs_lds != lds || e->queue_wgs_opt != (e->s.n_opt > 0)) {
                int per_cu = 0, cus = 0;
                if (e->s.n_opt > 0) HIPC(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_node_queue<512, false, true>, 512, lds));
                else HIPC(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_node_queue<512, true>, 512, lds));
                e->queue_wgs_opt = e->s.n_opt > 0;
                HIPC(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device));
                e->queue_wgs = std::max(1, per_cu * cus);
                e->queue_wgs_lds = lds;
            }
            const char* t = getenv("JSLP_GROUP_MAX");
            group = (int)std::min<long long>(std::min<long long>(n_nodes, t ? group_max() : e->queue_wgs), max_slots);
        }
        rc = ensure_slots(e, group);
        if (rc) return rc;
    }
    rc = ensure_out(e, (size_t)n_nodes);  // laid out for ALL nodes: [states | rhs | rows]
    if (rc) return rc;
    if (e->timing) HIPC(hipEventRecord(e->ev_begin, s));
    // where the kernels leave the outcomes: the device staging buffer (copied group by group on the copy stream) or, zero-copy,
    // the pinned host buffer itself - the stores cross PCIe while the other workgroups compute
    DevState* o_states = e->d_states; double* o_rhs = e->d_rhs; int32_t* o_rows = e->d_rows;
    bool zc = zero_copy() != 0;
    unsigned* polled_flag = nullptr;  // set when the call ends by polling a completion flag instead of synchronising the streams
    unsigned polled_seq = 0;
    if (dev_out) { o_states = e->dev_states; o_rhs = e->dev_rhs; o_rows = e->dev_rows; zc = false; }
    if (zc) {
        void *ps = nullptr, *pr = nullptr, *pw = nullptr;
        if (hipHostGetDevicePointer(&ps, e->h_states, 0) == hipSuccess && hipHostGetDevicePointer(&pr, e->h_rhs, 0) == hipSuccess &&
            hipHostGetDevicePointer(&pw, e->h_rows, 0) == hipSuccess) {
            o_states = static_cast<DevState*>(ps); o_rhs = static_cast<double*>(pr); o_rows = static_cast<int32_t*>(pw);
        } else {
            (void)hipGetLastError();
            zc = false;
        }
    }
    if (!zc && !dev_out && !e->copy_stream) {
        // created on first use, and only by the calls that copy their outcomes back (JSLP_ZERO_COPY=0, no mapped pinned memory): a stream costs a Solve of a
        // tiny model more than its pivots do
        HIPC(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
        HIPC(hipEventCreateWithFlags(&e->ev_group, hipEventDisableTiming));
    }
    if (const size_t lds = wglds_smem(e); wg && lds && node_queue() && n_nodes > group && checkpoint < 0 && e->has_save && e->slot0_synced &&
        group <= e->slots_synced && !e->timing && e->one_launch_nodes && wg_batch_threads() == 512) {
        Snapshot sn = root_snapshot(e);
        e->last_path = "workgroup";
        e->node_queue_launches += 1;
        // (an order only where it can matter: more nodes than resident workgroups -- a batch that starts all at once pays the order array's trip per node
        //  for nothing: the four-member pool's 604-node shares lost 6 % to it; JSLP_NODE_QUEUE_ORDER_FULL=1 hands the FULL read-back out longest-first too; its bound is
        //  the PCIe link and it measured 3.41-3.42 M against 3.52-3.54 M relaxations/s in call order: the heavy nodes' outcomes then leave in one burst)
        static const int order_full = [] { const char* t = getenv("JSLP_NODE_QUEUE_ORDER_FULL"); return t ? atoi(t) : 0; }();
        const bool pcie_bound = zc && !compact && (want_rhs || want_rows);
        const int32_t* order = (node_queue() == 2 && n_nodes > group && (order_full || !pcie_bound)) ? e->d_cut_order : (const int32_t*)nullptr;
        if (e->s.n_opt > 0)  // optional objectives: the OPT build (eager restores)
            hipLaunchKernelGGL((k_node_queue<512, false, true>), dim3(group), dim3(512), lds, s, e->s, sn, cu, (int)n_nodes, order, e->d_queue, check_cycles,
                               cap, (int)e->cap_rows, want_rhs ? o_rhs : nullptr, want_rows ? o_rows : nullptr, o_states, g_stride);
        else if (node_cow())
            hipLaunchKernelGGL((k_node_queue<512, true>), dim3(group), dim3(512), lds, s, e->s, sn, cu, (int)n_nodes, order, e->d_queue, check_cycles,
                               cap, (int)e->cap_rows, want_rhs ? o_rhs : nullptr, want_rows ? o_rows : nullptr, o_states, g_stride);
        else
            hipLaunchKernelGGL((k_node_queue<512, false>), dim3(group), dim3(512), lds, s, e->s, sn, cu, (int)n_nodes, order, e->d_queue, check_cycles,
                               cap, (int)e->cap_rows, want_rhs ? o_rhs : nullptr, want_rows ? o_rows : nullptr, o_states, g_stride);
        HIPC(hipGetLastError());
        if (!zc && !dev_out) {
            HIPC(hipMemcpyAsync(e->h_states, e->d_states, sizeof(DevState) * (size_t)n_nodes, hipMemcpyDeviceToHost, s));
            if (want_rhs) HIPC(hipMemcpyAsync(e->h_rhs, e->d_rhs, sizeof(double) * (size_t)n_nodes * row_stride, hipMemcpyDeviceToHost, s));
            if (want_rows) HIPC(hipMemcpyAsync(e->h_rows, e->d_rows, sizeof(int32_t) * (size_t)n_nodes * row_stride, hipMemcpyDeviceToHost, s));
        }
    } else
    for (int first = 0; first < n_nodes; first += group) {
        const int g = std::min(group, n_nodes - first);
        // slots already in sync with the snapshot: restore of the dirty rows, cuts, simplex and gather in ONE launch per
        // group (what a workgroup restores and cuts stays in its XCD's L2 for its own pivots)
        const bool one_launch = wg && g > 1 && checkpoint < 0 && e->has_save && e->slot0_synced && g <= e->slots_synced &&
                                !e->timing && e->one_launch_nodes && wg_batch_threads() == 512;
        if (one_launch) {
            Snapshot sn = root_snapshot(e);
            e->last_path = "workgroup";
            // a batch that leaves most CUs idle anyway (speculative batches of a real tree: <= 16 nodes) is served by latency, not
            // by occupancy: the 1024-thread shape of the single-node path, one workgroup per CU
            static const int small_1024 = getenv("JSLP_SMALL_BATCH_1024") ? atoi(getenv("JSLP_SMALL_BATCH_1024")) : JSLP_SMALL_BATCH_1024_DEFAULT;
            const size_t lds = wglds_smem(e);
            if (lds && e->s.n_opt > 0)  // optional objectives: the 1024-thread build only, whatever the batch size
                hipLaunchKernelGGL((k_node_lds<1024, true>), dim3(g), dim3(1024), lds, s, e->s, sn, cu, first, check_cycles, cap,
                                   (int)e->cap_rows, want_rhs ? o_rhs : nullptr, want_rows ? o_rows : nullptr, o_states,
                                   g_stride, first, (unsigned*)nullptr, 0u);
            else if (lds && g <= small_1024) {
                // a batch that is ONE group, its outcomes written straight into pinned memory: the last workgroup raises the completion
                // flag the host polls (below) instead of the two stream synchronisations that end the other shapes of this call
                unsigned* d_flag = nullptr;
                unsigned seq = 0;
                if (g == n_nodes && zc && !dev_out && !e->ext_states && batch_poll_on()) {
                    unsigned* h_flag = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(e->h_state) + sizeof(DevState));
                    void* flag_dev = nullptr;
                    if (hipHostGetDevicePointer(&flag_dev, h_flag, 0) == hipSuccess) {
                        d_flag = static_cast<unsigned*>(flag_dev);
                        seq = ++e->done_seq ? e->done_seq : ++e->done_seq;  // never 0 (the flag's initial value)
                        polled_flag = h_flag; polled_seq = seq;
                    } else {
                        (void)hipGetLastError();
                    }
                }
                // (round 6) copy-on-write start here too, as the queue kernel and the single-node call have it: a node READS what its slot's earlier nodes
                // dirtied from the root instead of restoring those rows first -- the eager restore was 8.6 k of the 78 k cycles of a node of an 8-node batch
                // (tools/wglds_timing.py one 8, debug build), in front of everything else; JSLP_NODE_COW_SMALL=0: eager restores
                static const int cow_small = getenv("JSLP_NODE_COW_SMALL") ? atoi(getenv("JSLP_NODE_COW_SMALL")) : 1;
                if (cow_small && node_cow())
                    hipLaunchKernelGGL((k_node_lds<1024, false, true>), dim3(g), dim3(1024), lds, s, e->s, sn, cu, first, check_cycles, cap,
                                       (int)e->cap_rows, want_rhs ? o_rhs : nullptr, want_rows ? o_rows : nullptr, o_states,
                                       g_stride, first, d_flag, seq, d_flag ? e->d_done_count : (int*)nullptr);
                else
                hipLaunchKernelGGL((k_node_lds<1024>), dim3(g), dim3(1024), lds, s, e->s, sn, cu, first, check_cycles, cap,
                                   (int)e->cap_rows, want_rhs ? o_rhs : nullptr, want_rows ? o_rows : nullptr, o_states,
                                   g_stride, first, d_flag, seq, d_flag ? e->d_done_count : (int*)nullptr);
            }
            else if (lds)
                hipLaunchKernelGGL((k_node_lds<512>), dim3(g), dim3(512), lds, s, e->s, sn, cu, first, check_cycles, cap,
                                   (int)e->cap_rows, want_rhs ? o_rhs : nullptr, want_rows ? o_rows : nullptr, o_states,
                                   g_stride, first, (unsigned*)nullptr, 0u);
            else
                hipLaunchKernelGGL((k_node_wg<512, 2048>), dim3(g), dim3(512), 0, s, e->s, sn, cu, first, check_cycles, cap,
                                   (int)e->cap_rows, want_rhs ? o_rhs : nullptr, want_rows ? o_rows : nullptr, o_states,
                                   g_stride, first, (unsigned*)nullptr, 0u);
            HIPC(hipGetLastError());
        } else {
        rc = enqueue_restore(e, 0, g, checkpoint);
        if (rc) return rc;
        hipLaunchKernelGGL(k_add_cuts, dim3(g), dim3(256), 0, s, e->s, cu, 0, first, (int)e->cap_rows);
        HIPC(hipGetLastError());
        if (wg) {
            e->last_path = "workgroup";
            // one node: the 1024-thread latency shape; a batch: smaller workgroups, more nodes in flight per CU
            const size_t lds = wglds_smem(e);
            const bool opt = lds && e->s.n_opt > 0;
            const int shape = (g == 1 || opt) ? 1024 : wg_batch_threads();
            if (opt)
                hipLaunchKernelGGL((k_simplex_lds<JSLP_WG_THREADS, true>), dim3(g), dim3(JSLP_WG_THREADS), lds, s, e->s, 0, check_cycles, cap, (int)e->cap_rows);
            else if (lds && shape == 1024)
                hipLaunchKernelGGL((k_simplex_lds<JSLP_WG_THREADS>), dim3(g), dim3(JSLP_WG_THREADS), lds, s, e->s, 0, check_cycles, cap, (int)e->cap_rows);
            else if (lds && shape == 512)
                hipLaunchKernelGGL((k_simplex_lds<512>), dim3(g), dim3(512), lds, s, e->s, 0, check_cycles, cap, (int)e->cap_rows);
            else if (shape == 256)
                hipLaunchKernelGGL((k_simplex_wg<256, 1024>), dim3(g), dim3(256), 0, s, e->s, 0, check_cycles, cap);
            else if (shape == 512)
                hipLaunchKernelGGL((k_simplex_wg<512, 2048>), dim3(g), dim3(512), 0, s, e->s, 0, check_cycles, cap);
            else
                hipLaunchKernelGGL((k_simplex_wg<JSLP_WG_THREADS, 4096>), dim3(g), dim3(JSLP_WG_THREADS), 0, s, e->s, 0, check_cycles, cap);
            HIPC(hipGetLastError());
        } else {
            // big tableau: the chip-wide kernels on slot 0 (g == 1)
            HIPC(hipMemcpyAsync(e->h_state, e->s.st, sizeof(DevState), hipMemcpyDeviceToHost, s));
            HIPC(hipStreamSynchronize(s));
            rc = state_error(*e->h_state);
            if (rc) return rc;
            rc = run_simplex(e, check_cycles);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(k_gather, dim3(g), dim3(256), 0, s, e->s, 0, want_rhs ? o_rhs : nullptr,
                           want_rows ? o_rows : nullptr, o_states, g_stride, first);
        HIPC(hipGetLastError());
        }  // !one_launch
        // this group's outcomes cross PCIe on the copy stream while the next group computes (the three regions of the
        // read-back buffer are laid out for all nodes, so a group is one contiguous slice of each)
        if (zc || dev_out) continue;
        HIPC(hipEventRecord(e->ev_group, s));
        HIPC(hipStreamWaitEvent(e->copy_stream, e->ev_group, 0));
        HIPC(hipMemcpyAsync(e->h_states + first, e->d_states + first, sizeof(DevState) * (size_t)g, hipMemcpyDeviceToHost, e->copy_stream));
        if (want_rhs)
            HIPC(hipMemcpyAsync(e->h_rhs + (size_t)first * row_stride, e->d_rhs + (size_t)first * row_stride,
                                sizeof(double) * (size_t)g * row_stride, hipMemcpyDeviceToHost, e->copy_stream));
        if (want_rows)
            HIPC(hipMemcpyAsync(e->h_rows + (size_t)first * row_stride, e->d_rows + (size_t)first * row_stride,
                                sizeof(int32_t) * (size_t)g * row_stride, hipMemcpyDeviceToHost, e->copy_stream));
    }
    if (e->timing && wg) HIPC(hipEventRecord(e->ev_end, s));
    bool arrived = false;
    if (polled_flag) {  // the small batch's last workgroup ends with a system-scope release store of `polled_seq` into pinned memory
        unsigned spins = 0;
        const auto t_begin = std::chrono::steady_clock::now();
        while (!(arrived = (__atomic_load_n(polled_flag, __ATOMIC_ACQUIRE) == polled_seq))) {
            if ((++spins & 0x3fffu) == 0 &&
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() > 5.0) break;
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        // JSLP_DEBUG_STALL=<ms> (diagnosis): a dependent batch normally takes ~70 us; say on stderr when one took longer than
        // that many milliseconds and whether the time went in front of the poll (cut-list staging, the launch call) or in it (the GPU side)
        static const double stall_ms = [] { const char* t = getenv("JSLP_DEBUG_STALL"); return t ? atof(t) : 0.0; }();
        if (stall_ms > 0) {
            const auto t_end = std::chrono::steady_clock::now();
            const double host_ms = std::chrono::duration<double, std::milli>(t_begin - t_enter).count(), poll_ms = std::chrono::duration<double, std::milli>(t_end - t_begin).count();
            if (host_ms + poll_ms > stall_ms)
                fprintf(stderr, "[jslp] stall: a %d-node polled batch took %.3f ms in front of the poll (staging + launch) and %.3f ms polling (%u spins)\n", (int)n_nodes, host_ms, poll_ms, spins);
        }
    }
    if (!arrived) {  // every other shape -- and a polled batch that did not show up in 5 s: let the runtime report the fault
        const auto t_sync = std::chrono::steady_clock::now();
        if (e->copy_stream) HIPC(hipStreamSynchronize(e->copy_stream));
        HIPC(hipStreamSynchronize(s));
        static const double stall_ms2 = [] { const char* t = getenv("JSLP_DEBUG_STALL"); return t ? atof(t) : 0.0; }();
        if (stall_ms2 > 0) {  // (the same diagnostic for the synchronising shapes: enqueue time vs time inside the two stream synchronisations)
            const auto t_end = std::chrono::steady_clock::now();
            const double host_ms = std::chrono::duration<double, std::milli>(t_sync - t_enter).count(), sync_ms = std::chrono::duration<double, std::milli>(t_end - t_sync).count();
            if (host_ms + sync_ms > stall_ms2)
                fprintf(stderr, "[jslp] stall: a %d-node batch took %.3f ms to enqueue (staging + launches) and %.3f ms in hipStreamSynchronize\n", (int)n_nodes, host_ms, sync_ms);
        }
    }
    if (wg && e->timing) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e->ev_begin, e->ev_end) == hipSuccess) e->total_ms += ms;
    }
    if (e->counting) e->wc.cut_rows += cut_offsets[n_nodes];
    if (dev_out) {
        // the outcomes stay on the device (the caller converts the gathered records with jslp_engine_results_from_states), but the
        // engine's own bookkeeping must not depend on where they went: the 128-byte records of THIS call come back (n x 128 B) so
        // that a node that ended in an error clears the slot-sync flags (later one-launch / copy-on-write batches must not start
        // from slots it left dirty), the work counters see the pivots, and the engine's evaluation is the last node's
        HIPC(hipMemcpy(e->h_states, e->dev_states, sizeof(DevState) * (size_t)n_nodes, hipMemcpyDeviceToHost));
        e->dev_prev_evaluation = prev_eval;  // what results_from_states hands a node that reaches no optimum (as the host path does)
        e->dev_prev_valid = 1;
        for (int i = 0; i < n_nodes; i++) {
            const DevState& st = e->h_states[i];
            rc = state_error(st);
            if (rc) { e->slot0_synced = 0; e->slots_synced = 0; return rc; }
            account(e, st, 1);
            if (i == n_nodes - 1) {
                jslp_simplex_result last;
                DevState rec = st;
                rec.hist_n = 0;
                rc = fill_result(e, rec, 0, prev_eval, &last, &e->evaluation);
                if (rc) return rc;
            }
        }
        return JSLP_OK;
    }
    for (int i = 0; i < n_nodes; i++) {
        DevState st = e->h_states[i];
        rc = state_error(st);
        if (rc) { e->slot0_synced = 0; e->slots_synced = 0; return rc; }
        account(e, st, 1);
        if (st.cycle_phase && wg && n_nodes > group) {
            // the cycle message is rebuilt from the slot's history, which later groups have reused: report the hit
            // (flags are exact) without the [start, length] detail
            st.hist_n = 0;
        }
        double ev;
        rc = fill_result(e, st, wg ? i % group : 0, prev_eval, &out[i], &ev);
        if (rc) return rc;
        if (i == n_nodes - 1) e->evaluation = ev;
        if (!pinned && !compact) {
            const size_t n_out = (size_t)st.H;
            if (rhs) memcpy(rhs + (size_t)i * out_stride, e->h_rhs + (size_t)i * row_stride, sizeof(double) * n_out);
            if (var_index_by_row)
                memcpy(var_index_by_row + (size_t)i * out_stride, e->h_rows + (size_t)i * row_stride, sizeof(int32_t) * n_out);
        }
    }
    if (!pinned && compact) {  // same layout on both sides: one copy each (per-node copies cost more than the kernel's share of a node)
        if (rhs) memcpy(rhs, e->h_rhs, sizeof(double) * (size_t)n_nodes * row_stride);
        if (var_index_by_row) memcpy(var_index_by_row, e->h_rows, sizeof(int32_t) * (size_t)n_nodes * row_stride);
    }
    return JSLP_OK;
}

extern "C" int jslp_engine_relax_batch(jslp_engine* e, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                                       const int32_t* var_index, const double* value, int check_cycles,
                                       jslp_simplex_result* out, double* rhs, int32_t* var_index_by_row,
                                       int32_t out_stride) {
    return relax_batch_impl(e, n_nodes, cut_offsets, type, var_index, value, check_cycles, out, rhs, var_index_by_row,
                            out_stride, 0, rhs != nullptr, var_index_by_row != nullptr);
}

// ---- outcomes left in DEVICE memory of the caller (the N > 1 process path: they are the input of an RCCL all-gather) -------------
extern "C" int32_t jslp_engine_state_record_bytes(void) { return (int32_t)sizeof(DevState); }

extern "C" int jslp_engine_relax_batch_device(jslp_engine* e, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                                              const int32_t* var_index, const double* value, int check_cycles, void* d_states,
                                              double* d_rhs, int32_t* d_rows, int32_t row_stride) {
    if (!e || !d_states || !d_rhs || !d_rows) return fail(JSLP_ERR_ARG, "relax_batch_device: null pointer");
    if (row_stride < e->cap_rows) return fail(JSLP_ERR_ARG, "relax_batch_device: row_stride < row capacity");
    e->dev_states = static_cast<DevState*>(d_states); e->dev_rhs = d_rhs; e->dev_rows = d_rows; e->dev_stride = row_stride;
    const int rc = relax_batch_impl(e, n_nodes, cut_offsets, type, var_index, value, check_cycles, nullptr, nullptr, nullptr, 0, 1, 1, 1);
    e->dev_states = nullptr; e->dev_rhs = nullptr; e->dev_rows = nullptr; e->dev_stride = 0;
    return rc;
}

// `states`: n_nodes raw records in HOST memory (wherever they were evaluated: this rank's own or another rank's, after the
// exchange) -> the result structs every other entry point returns.  A record carries no pivot history: the [start, length] detail of
// a detected cycle is reported as 0 (the flags and the phase are exact).
extern "C" int jslp_engine_results_from_states(jslp_engine* e, const void* states, int32_t n_nodes, jslp_simplex_result* out) {
    if (!e || !states || !out || n_nodes < 0) return fail(JSLP_ERR_ARG, "results_from_states: bad arguments");
    const DevState* st = static_cast<const DevState*>(states);
    for (int32_t i = 0; i < n_nodes; i++) {
        DevState rec = st[i];
        int rc = state_error(rec);
        if (rc) return rc;
        rec.hist_n = 0;
        double ev;
        rc = fill_result(e, rec, 0, e->dev_prev_valid ? e->dev_prev_evaluation : e->evaluation, &out[i], &ev);
        if (rc) return rc;
    }
    return JSLP_OK;
}

extern "C" int jslp_engine_relax_batch_pinned(jslp_engine* e, int32_t n_nodes, const int32_t* cut_offsets,
                                              const int8_t* type, const int32_t* var_index, const double* value,
                                              int check_cycles, jslp_simplex_result* out, const double** rhs,
                                              const int32_t** var_index_by_row, int32_t* out_stride) {
    if (!e) return fail(JSLP_ERR_ARG, "relax_batch_pinned: null engine");
    int rc = relax_batch_impl(e, n_nodes, cut_offsets, type, var_index, value, check_cycles, out, nullptr, nullptr, 0, 1,
                              rhs != nullptr, var_index_by_row != nullptr);
    if (rc) return rc;
    if (rhs) *rhs = n_nodes > 0 ? e->h_rhs : nullptr;
    if (var_index_by_row) *var_index_by_row = n_nodes > 0 ? e->h_rows : nullptr;
    if (out_stride) *out_stride = (int32_t)out_stride_of(e);
    return JSLP_OK;
}

extern "C" int jslp_engine_relax_from(jslp_engine* e, int32_t checkpoint, int32_t n_nodes, const int32_t* cut_offsets,
                                      const int8_t* type, const int32_t* var_index, const double* value, int check_cycles,
                                      jslp_simplex_result* out, double* rhs, int32_t* var_index_by_row,
                                      int32_t out_stride) {
    return relax_batch_impl(e, n_nodes, cut_offsets, type, var_index, value, check_cycles, out, rhs, var_index_by_row,
                            out_stride, 0, rhs != nullptr, var_index_by_row != nullptr, checkpoint < 0 ? -1 : checkpoint);
}

extern "C" int jslp_engine_relax(jslp_engine* e, int32_t n_cuts, const int8_t* type, const int32_t* var_index,
                                 const double* value, int check_cycles, jslp_simplex_result* out, double* rhs,
                                 int32_t* var_index_by_row) {
    if (!e) return fail(JSLP_ERR_ARG, "relax: null engine");
    if (n_cuts < 0) return fail(JSLP_ERR_ARG, "relax: negative cut count");
    const int32_t offs[2] = {0, n_cuts};
    return jslp_engine_relax_batch(e, 1, offs, type, var_index, value, check_cycles, out, rhs, var_index_by_row,
                                   e->cap_rows);
}

// ---- compact read-back -----------------------------------------------------------------------------------------------
extern "C" int jslp_engine_set_watched_variables(jslp_engine* e, const int32_t* var_indexes, int32_t n) {
    if (!e || !e->uploaded) return fail(JSLP_ERR_STATE, "set_watched_variables before upload");
    if (n < 0 || (n > 0 && !var_indexes)) return fail(JSLP_ERR_ARG, "set_watched_variables: bad arguments");
    for (int32_t i = 0; i < n; i++)
        if (var_indexes[i] < 0 || var_indexes[i] >= e->n_idx) return fail(JSLP_ERR_ARG, "set_watched_variables: index out of range");
    HIPC(hipSetDevice(e->device));
    HIPC(hipStreamSynchronize(e->stream));
    hipFree(e->d_watch);
    e->d_watch = nullptr; e->n_watch = 0; e->s.watch = nullptr; e->s.n_watch = 0; e->s.watch_pos = nullptr;
    if (n > 0) {
        // [the list | variable index -> position in the list] (the second half only when no variable is listed twice)
        std::vector<int32_t> pos((size_t)e->n_idx, -1);
        bool unique = true;
        for (int32_t i = 0; i < n; i++) {
            if (pos[var_indexes[i]] >= 0) unique = false;
            pos[var_indexes[i]] = i;
        }
        HIPC(hipMalloc(&e->d_watch, sizeof(int32_t) * ((size_t)n + (size_t)e->n_idx)));
        HIPC(hipMemcpy(e->d_watch, var_indexes, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice));
        HIPC(hipMemcpy(e->d_watch + n, pos.data(), sizeof(int32_t) * (size_t)e->n_idx, hipMemcpyHostToDevice));
        e->n_watch = n; e->s.watch = e->d_watch; e->s.n_watch = n;
        e->s.watch_pos = unique ? e->d_watch + n : nullptr;
    }
    return JSLP_OK;
}

extern "C" int jslp_engine_relax_watched(jslp_engine* e, int32_t n_cuts, const int8_t* type, const int32_t* var_index,
                                         const double* value, int check_cycles, jslp_simplex_result* out,
                                         int32_t* watched_row, double* watched_value) {
    if (!e) return fail(JSLP_ERR_ARG, "relax_watched: null engine");
    if (n_cuts < 0) return fail(JSLP_ERR_ARG, "relax_watched: negative cut count");
    const int32_t offs[2] = {0, n_cuts};
    return relax_batch_impl(e, 1, offs, type, var_index, value, check_cycles, out, watched_value, watched_row, e->cap_rows, 0,
                            watched_value != nullptr, watched_row != nullptr, -1, 1);
}

extern "C" int jslp_engine_relax_batch_watched(jslp_engine* e, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                                               const int32_t* var_index, const double* value, int check_cycles,
                                               jslp_simplex_result* out, int32_t* watched_row, double* watched_value) {
    if (!e) return fail(JSLP_ERR_ARG, "relax_batch_watched: null engine");
    return relax_batch_impl(e, n_nodes, cut_offsets, type, var_index, value, check_cycles, out, watched_value, watched_row, e->cap_rows, 0,
                            watched_value != nullptr, watched_row != nullptr, -1, 1);
}

extern "C" int jslp_engine_relax_batch_watched_pinned(jslp_engine* e, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                                                      const int32_t* var_index, const double* value, int check_cycles,
                                                      jslp_simplex_result* out, const int32_t** watched_row,
                                                      const double** watched_value) {
    if (!e) return fail(JSLP_ERR_ARG, "relax_batch_watched_pinned: null engine");
    int rc = relax_batch_impl(e, n_nodes, cut_offsets, type, var_index, value, check_cycles, out, nullptr, nullptr, 0, 1,
                              watched_value != nullptr, watched_row != nullptr, -1, 1);
    if (rc) return rc;
    if (watched_row) *watched_row = n_nodes > 0 ? e->h_rows : nullptr;
    if (watched_value) *watched_value = n_nodes > 0 ? e->h_rhs : nullptr;
    return JSLP_OK;
}

// ---- work counters ---------------------------------------------------------------------------------------------------
// the compact read-back left in the caller's DEVICE memory: the exchange payload of the multi-process path (sharding.py)
extern "C" int jslp_engine_relax_batch_watched_device(jslp_engine* e, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                                                      const int32_t* var_index, const double* value, int check_cycles, void* d_states,
                                                      int32_t* d_watched_row, double* d_watched_value) {
    if (!e || !d_states || !d_watched_row || !d_watched_value) return fail(JSLP_ERR_ARG, "relax_batch_watched_device: null pointer");
    e->dev_states = static_cast<DevState*>(d_states); e->dev_rhs = d_watched_value; e->dev_rows = d_watched_row; e->dev_stride = e->n_watch;
    const int rc = relax_batch_impl(e, n_nodes, cut_offsets, type, var_index, value, check_cycles, nullptr, nullptr, nullptr, 0, 1, 1, 1, -1, 1);
    e->dev_states = nullptr; e->dev_rhs = nullptr; e->dev_rows = nullptr; e->dev_stride = 0;
    return rc;
}

extern "C" int32_t jslp_engine_watched_count(const jslp_engine* e) { return e ? e->n_watch : 0; }

extern "C" int jslp_engine_set_counting(jslp_engine* e, int enabled) {
    if (!e) return fail(JSLP_ERR_ARG, "set_counting: null engine");
    HIPC(hipSetDevice(e->device));
    HIPC(hipStreamSynchronize(e->stream));
    if (!e->d_cnt) HIPC(hipMalloc(&e->d_cnt, sizeof(cnt_t) * CNT_ALLOC));
    HIPC(hipMemset(e->d_cnt, 0, sizeof(cnt_t) * CNT_ALLOC));
    e->counting = enabled ? 1 : 0;
    e->s.cnt = enabled ? e->d_cnt : nullptr;
    e->wc = jslp_work_counters{};
    e->resident_fallbacks = e->resident_handovers = e->resident_launches = e->resident_refusals = e->node_queue_launches = 0;
    e->resident_fetch_retries = 0;
    return JSLP_OK;
}

extern "C" int jslp_engine_get_counters(jslp_engine* e, jslp_work_counters* out) {
    if (!e || !out) return fail(JSLP_ERR_ARG, "get_counters: null pointer");
    *out = e->wc;
    out->resident_aborts = e->resident_fallbacks;
    out->resident_handovers = e->resident_handovers;
    out->resident_launches = e->resident_launches;
    out->resident_refusals = e->resident_refusals;
    out->node_queue_launches = e->node_queue_launches;
    out->resident_fetch_retries = e->resident_fetch_retries;
    if (e->d_cnt) {
        HIPC(hipSetDevice(e->device));
        HIPC(hipStreamSynchronize(e->stream));
        cnt_t c[CNT_ALLOC];
        HIPC(hipMemcpy(c, e->d_cnt, sizeof c, hipMemcpyDeviceToHost));
#ifdef JSLP_DEBUG_WGLDS
        {   // cycle accumulators of the LDS one-workgroup kernels (thread 0 of every workgroup, s_memtime), per section
            static const char* names[] = {"restore rows", "restore maps", "cuts", "begin + LDS load", "phase-1 row", "pricing", "column gather + ratio test",
                                          "cycle check", "gate + pivot row", "map swap + compaction", "row updates", "epilogue", "read-back", "phase-1 column"};
            const double piv = (double)std::max<long long>(1, out->pivots), nodes = (double)std::max<long long>(1, out->simplex_calls);
            fprintf(stderr, "[wglds cycles] per node (%lld nodes, %lld pivots):", (long long)out->simplex_calls, (long long)out->pivots);
            double tot = 0;
            for (int i = 0; i < 14; i++) tot += (double)c[CNT_DBG + i];
            for (int i = 0; i < 14; i++)
                fprintf(stderr, "\n  %-28s %10.0f cycles/node %9.0f cycles/pivot %5.1f %%", names[i], c[CNT_DBG + i] / nodes, c[CNT_DBG + i] / piv, 100.0 * c[CNT_DBG + i] / std::max(1.0, tot));
            fprintf(stderr, "\n  total %.0f cycles/node\n", tot / nodes);
            static const char* micro[] = {"block_min_ki", "__syncthreads", "LDS scan of H + barrier", "strided column gather + barrier", "coalesced row load + barrier",
                                          "coalesced row store + barrier"};
            for (int i = 0; i < 6; i++) fprintf(stderr, "  micro: %-34s %8.0f cycles each\n", micro[i], c[CNT_DBG + 14 + i] / nodes / 16.0);
        }
#endif
        out->gated_cells += (int64_t)c[CNT_CELLS];
        out->gated_rows += (int64_t)c[CNT_ROWS];
        out->restored_rows += (int64_t)c[CNT_RESTORED];
    }
    return JSLP_OK;
}

// ---- device pool -------------------------------------------------------------------------------------------
// One host thread per additional member: the caller stays single-threaded and synchronous, the fan-out lives in here.
// One host thread per further pool member.  A batch call hands every member its share and waits for all of them; with a condition variable
// on both sides the hand-off cost ~250 us per call on the bench box (four members: three wake-ups to start, three to finish) -- most of what a
// 0.6 ms batch call takes.  Round 5: both sides SPIN first (a worker for ~0.2 ms after its last job, the caller while its members run) and fall
// back to the condition variable only when nothing arrives: calls that follow one another -- a tree's batches, the bench's repeated batch --
// never sleep.
struct PoolWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    std::atomic<unsigned> submitted{0}, done{0};
    std::atomic<int> asleep{0}, quit{0};
    int rc = 0;
    char err[512] = {0};
    static void relax_cpu() {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    // (the sleep / wake handshake is a Dekker pattern -- the worker stores `asleep` and then loads `submitted`, submit() bumps
    //  `submitted` and then loads `asleep` -- so every access that takes part in it is seq_cst: with acquire loads in the predicate the
    //  C++ memory model allows the store and the load to pass each other and a wake-up to be lost; both sides spin for a bounded TIME
    //  -- ~0.5 ms in the worker, ~0.2 ms in the caller -- not for a number of iterations, before they give the core back)
    static bool spin_for_us(const std::chrono::steady_clock::time_point& t0, unsigned& spins, double us) {
        relax_cpu();
        return (++spins & 0xffu) != 0 || std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < us;
    }
    void loop() {
        unsigned seen = 0;
        for (;;) {
            unsigned spins = 0;
            auto t0 = std::chrono::steady_clock::now();
            while (submitted.load(std::memory_order_seq_cst) == seen && !quit.load(std::memory_order_seq_cst)) {
                if (spin_for_us(t0, spins, 500.0)) continue;
                std::unique_lock<std::mutex> lk(mu);
                asleep.store(1, std::memory_order_seq_cst);
                cv.wait(lk, [&] { return submitted.load(std::memory_order_seq_cst) != seen || quit.load(std::memory_order_seq_cst) != 0; });
                asleep.store(0, std::memory_order_seq_cst);
                spins = 0;
                t0 = std::chrono::steady_clock::now();
            }
            if (quit.load(std::memory_order_acquire)) return;
            seen = submitted.load(std::memory_order_acquire);
            g_err[0] = 0;
            const int r = job();
            rc = r;
            snprintf(err, sizeof err, "%s", g_err);
            done.store(seen, std::memory_order_release);
        }
    }
    void submit(std::function<int()> j) {
        job = std::move(j);  // (the worker reads it only behind the acquire of `submitted`; the previous job is done: wait() returned)
        submitted.fetch_add(1, std::memory_order_seq_cst);
        if (asleep.load(std::memory_order_seq_cst)) {
            std::lock_guard<std::mutex> lk(mu);
            cv.notify_all();
        }
    }
    int wait() {
        unsigned spins = 0;
        const auto t0 = std::chrono::steady_clock::now();
        while (done.load(std::memory_order_acquire) != submitted.load(std::memory_order_acquire)) {
            if (spin_for_us(t0, spins, 200.0)) continue;
            std::this_thread::sleep_for(std::chrono::microseconds(20));  // (a long job -- a big batch, a root fan-out over a slow link: stop burning the core)
        }
        return rc;
    }
    void stop() {
        wait();
        quit.store(1, std::memory_order_seq_cst);
        {
            std::lock_guard<std::mutex> lk(mu);
            cv.notify_all();
        }
        if (th.joinable()) th.join();
    }
};

struct jslp_pool {
    std::vector<jslp_engine*> members;  // members[0] = the primary (not owned)
    std::vector<PoolWorker*> workers;   // workers[i] drives members[i] (i >= 1); the primary runs on the calling thread
    unsigned long long synced_seq = ~0ull;
    int synced = 0;
    char* h_out = nullptr; size_t h_out_bytes = 0;  // ONE pinned (portable) read-back buffer: [states | rhs | rows] for all nodes
    std::vector<std::vector<int32_t>> offs;         // per member: its cut offsets rebased to 0
};

extern "C" int jslp_pool_size(const jslp_pool* p) { return p ? (int)p->members.size() : 0; }

extern "C" void jslp_pool_destroy(jslp_pool* p) {
    if (!p) return;
    for (size_t i = 1; i < p->workers.size(); i++) {
        PoolWorker* w = p->workers[i];
        if (!w) continue;
        w->stop();
        delete w;
    }
    for (size_t i = 1; i < p->members.size(); i++) jslp_engine_destroy(p->members[i]);
    if (p->h_out) hipHostFree(p->h_out);
    delete p;
}

extern "C" int jslp_pool_create(jslp_pool** out, jslp_engine* primary, const int32_t* devices, int32_t n_devices) {
    if (!out || !primary || !devices || n_devices < 1) return fail(JSLP_ERR_ARG, "pool_create: bad arguments");
    if (devices[0] != primary->device) return fail(JSLP_ERR_ARG, "pool_create: devices[0] must be the primary engine's device");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(JSLP_ERR_DEVICE, "pool_create: no HIP device visible");
    for (int32_t i = 0; i < n_devices; i++)
        if (devices[i] < 0 || devices[i] >= ndev) return fail(JSLP_ERR_ARG, "pool_create: device ordinal out of range");
    jslp_pool* p = new jslp_pool();
    p->members.push_back(primary);
    p->workers.push_back(nullptr);
    for (int32_t i = 1; i < n_devices; i++) {
        jslp_engine* m = nullptr;
        int rc = jslp_engine_create(&m, devices[i], primary->H0, primary->W, primary->cap_rows, primary->precision);
        if (rc) { jslp_pool_destroy(p); return rc; }
        p->members.push_back(m);
        // The root fan-out (pool_adopt_root) is a set of peer copies primary -> member: direct xGMI transfers need peer access.
        // A platform that refuses it would still run -- the runtime stages peer copies through host memory -- at a fraction of
        // the bandwidth and silently: the pool fails LOUDLY instead (JSLP_POOL_ALLOW_STAGED=1 accepts the staged copies).
        // JSLP_TEST_PEER_REFUSED=1 (tests) makes every member look refused, virtual devices included.
        static const bool test_refused = getenv("JSLP_TEST_PEER_REFUSED") && atoi(getenv("JSLP_TEST_PEER_REFUSED")) != 0;
        static const bool allow_staged = getenv("JSLP_POOL_ALLOW_STAGED") && atoi(getenv("JSLP_POOL_ALLOW_STAGED")) != 0;
        if (devices[i] != primary->device || test_refused) {
            int can = 0;
            bool ok = !test_refused && hipDeviceCanAccessPeer(&can, devices[i], primary->device) == hipSuccess && can;
            if (ok) {
                hipSetDevice(devices[i]);
                const hipError_t pe = hipDeviceEnablePeerAccess(primary->device, 0);
                ok = pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled;
            }
            (void)hipGetLastError();
            if (!ok && !allow_staged) {
                hipSetDevice(primary->device);
                jslp_pool_destroy(p);  // members created so far, their worker threads; never the primary
                char msg[160];
                snprintf(msg, sizeof msg, "pool_create: device %d cannot access device %d's memory (peer access refused); "
                         "JSLP_POOL_ALLOW_STAGED=1 accepts host-staged copies", (int)devices[i], (int)primary->device);
                return fail(JSLP_ERR_DEVICE, msg);
            }
        }
        PoolWorker* w = new PoolWorker();
        w->th = std::thread([w] { w->loop(); });
        p->workers.push_back(w);
    }
    hipSetDevice(primary->device);
    p->offs.resize(p->members.size());
    *out = p;
    return JSLP_OK;
}

// one member adopts the primary's saved root: peer copies of the snapshot + a state fix-up + restore()
static int pool_adopt_root(jslp_engine* m, const jslp_engine* src, int s_H, int s_lei) {
    HIPC(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    if (m->n_opt != src->n_opt) {  // optional objectives travel with the root (backup.ts:37-43)
        HIPC(hipStreamSynchronize(s));
        hipFree(m->s.oo); hipFree(m->snap_oo);
        m->s.oo = nullptr; m->snap_oo = nullptr;
        m->n_opt = src->n_opt; m->s.n_opt = src->n_opt; m->s.oo_stride = (long long)src->n_opt * m->ld;
        if (src->n_opt > 0) {
            const size_t per = (size_t)m->s.oo_stride;
            HIPC(hipMalloc(&m->s.oo, sizeof(double) * per * std::max(1, m->n_slots)));
            HIPC(hipMalloc(&m->snap_oo, sizeof(double) * per));
            HIPC(hipMemsetAsync(m->s.oo, 0, sizeof(double) * per * std::max(1, m->n_slots), s));
        }
    }
    const int sd = src->device, dd = m->device;
    HIPC(hipMemcpyPeerAsync(m->snap_A, dd, src->snap_A, sd, sizeof(double) * (size_t)s_H * m->ld, s));
    HIPC(hipMemcpyPeerAsync(m->snap_rhs, dd, src->snap_rhs, sd, sizeof(double) * (size_t)s_H, s));
    HIPC(hipMemcpyPeerAsync(m->snap_vibr, dd, src->snap_vibr, sd, sizeof(int32_t) * (size_t)s_H, s));
    HIPC(hipMemcpyPeerAsync(m->snap_vibc, dd, src->snap_vibc, sd, sizeof(int32_t) * (size_t)m->W, s));
    HIPC(hipMemcpyPeerAsync(m->snap_rbv, dd, src->snap_rbv, sd, sizeof(int32_t) * (size_t)m->n_idx, s));
    HIPC(hipMemcpyPeerAsync(m->snap_cbv, dd, src->snap_cbv, sd, sizeof(int32_t) * (size_t)m->n_idx, s));
    HIPC(hipMemcpyPeerAsync(m->d_unr, dd, src->d_unr, sd, (size_t)m->n_idx, s));
    HIPC(hipMemcpyPeerAsync(m->d_isint, dd, src->d_isint, sd, (size_t)m->n_idx, s));
    if (src->n_opt > 0) HIPC(hipMemcpyPeerAsync(m->snap_oo, dd, src->snap_oo, sd, sizeof(double) * (size_t)m->s.oo_stride, s));
    hipLaunchKernelGGL(k_adopt_root, dim3(1), dim3(1), 0, s, m->s, s_H, s_lei);
    launch_snapshot_transpose(m);
    HIPC(hipGetLastError());
    m->uploaded = 1;
    m->has_save = 1;
    m->root_seq += 1;
    m->n_unr = src->n_unr;
    m->s.has_unr = src->n_unr > 0 ? 1 : 0;
    m->evaluation = src->evaluation;
    m->nnz = src->nnz;
    m->max_uploaded_idx = src->max_uploaded_idx;
    m->slot0_synced = 0;
    m->slots_synced = 0;
    drop_checkpoints(m, 0);
    int rc = enqueue_restore(m, 0, 1);  // the member's live tableau = the root
    if (rc) return rc;
    HIPC(hipStreamSynchronize(s));
    return JSLP_OK;
}

static int pool_join(jslp_pool* p, int rc0) {  // wait for every worker; the first failure wins (its text becomes ours)
    int rc = rc0;
    for (size_t i = 1; i < p->workers.size(); i++) {
        const int r = p->workers[i]->wait();
        if (r && !rc) { rc = r; snprintf(g_err, sizeof g_err, "%s", p->workers[i]->err); }
    }
    return rc;
}

extern "C" int jslp_pool_sync_root(jslp_pool* p) {
    if (!p) return fail(JSLP_ERR_ARG, "pool_sync_root: null pool");
    jslp_engine* e = p->members[0];
    if (!e->uploaded || !e->has_save) return fail(JSLP_ERR_STATE, "pool_sync_root: the primary has no saved root (save() first)");
    HIPC(hipSetDevice(e->device));
    HIPC(hipStreamSynchronize(e->stream));
    DevState st;
    HIPC(hipMemcpy(&st, e->s.st, sizeof st, hipMemcpyDeviceToHost));
    const int s_H = st.s_H, s_lei = st.s_last_element_index;
    for (size_t i = 1; i < p->members.size(); i++) {
        jslp_engine* m = p->members[i];
        p->workers[i]->submit([m, e, s_H, s_lei] { return pool_adopt_root(m, e, s_H, s_lei); });
    }
    int rc = pool_join(p, JSLP_OK);
    hipSetDevice(e->device);
    if (rc) return rc;
    p->synced_seq = e->root_seq;
    p->synced = 1;
    return JSLP_OK;
}

// compact = 1: the watched variables' rows / RHS cells only (rhs / vibr then hold n_watch entries per node, out_stride is ignored)
static int pool_relax(jslp_pool* p, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type, const int32_t* var_index,
                      const double* value, int check_cycles, jslp_simplex_result* out, double* rhs, int32_t* vibr,
                      int32_t out_stride, int want_rhs, int want_rows, int compact = 0) {
    if (!p || n_nodes < 0 || !cut_offsets || !out) return fail(JSLP_ERR_ARG, "pool_relax_batch: null pointer");
    jslp_engine* e = p->members[0];
    if (!e->uploaded || !e->has_save) return fail(JSLP_ERR_STATE, "pool_relax_batch: the primary has no saved root (save() first)");
    if (!compact && (rhs || vibr) && out_stride < e->cap_rows) return fail(JSLP_ERR_ARG, "pool_relax_batch: out_stride < row capacity");
    if (compact) {
        if (e->n_watch <= 0) return fail(JSLP_ERR_ARG, "pool_relax_batch_watched: after jslp_pool_set_watched_variables");
        for (jslp_engine* m : p->members)
            if (m->n_watch != e->n_watch) return fail(JSLP_ERR_STATE, "pool_relax_batch_watched: the members' watched variables differ from the primary's (jslp_pool_set_watched_variables sets them all)");
    }
    if (n_nodes == 0) return JSLP_OK;
    if (cut_offsets[0] != 0) return fail(JSLP_ERR_ARG, "cuts: cut_offsets[0] must be 0");
    for (int32_t i = 0; i < n_nodes; i++)
        if (cut_offsets[i + 1] < cut_offsets[i]) return fail(JSLP_ERR_ARG, "cuts: cut_offsets must not decrease");
    int rc;
    if (!p->synced || p->synced_seq != e->root_seq) {  // the primary saved a new root since the last fan-out
        rc = jslp_pool_sync_root(p);
        if (rc) return rc;
    }
    // ONE pinned buffer for every member's outcomes, laid out for all nodes: [states | rhs | rows]
    const size_t cap = compact ? (size_t)e->n_watch : (size_t)e->cap_rows;  // entries per node in the shared buffer
    const size_t need = (size_t)n_nodes * (sizeof(DevState) + cap * 12);
    if (need > p->h_out_bytes) {
        if (p->h_out) hipHostFree(p->h_out);
        p->h_out = nullptr; p->h_out_bytes = 0;
        const size_t bytes = std::max<size_t>(need, (size_t)16 * (sizeof(DevState) + cap * 12));
        HIPC(hipHostMalloc(&p->h_out, bytes, hipHostMallocPortable));
        p->h_out_bytes = bytes;
    }
    DevState* g_states = reinterpret_cast<DevState*>(p->h_out);
    double* g_rhs = reinterpret_cast<double*>(p->h_out + (size_t)n_nodes * sizeof(DevState));
    int32_t* g_rows = reinterpret_cast<int32_t*>(p->h_out + (size_t)n_nodes * (sizeof(DevState) + cap * 8));
    const int M = (int)p->members.size();
    // JSLP_DEBUG_POOL=1 (diagnosis, tools/pool_handoff.py): per call, when every member's job started and ended relative to the call's entry -- what of a
    // pool call is thread hand-off (the start delays, the join after the last end) and what is the members' own work
    static const int dbg_pool = [] { const char* t = getenv("JSLP_DEBUG_POOL"); return t ? atoi(t) : 0; }();
    const auto t_call = std::chrono::steady_clock::now();
    static double dbg_t[2][64];
    auto job = [=](int mi) -> int {
        jslp_engine* m = p->members[mi];
        const int first = (int)((long long)n_nodes * mi / M), last = (int)((long long)n_nodes * (mi + 1) / M), cnt = last - first;
        struct Stamp {
            int on, mi; std::chrono::steady_clock::time_point t0;
            Stamp(int on_, int mi_, std::chrono::steady_clock::time_point t) : on(on_), mi(mi_), t0(t) { if (on && mi < 64) dbg_t[0][mi] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
            ~Stamp() { if (on && mi < 64) dbg_t[1][mi] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
        } stamp(dbg_pool, mi, t_call);
        if (cnt <= 0) return JSLP_OK;
        std::vector<int32_t>& o = p->offs[mi];
        o.resize((size_t)cnt + 1);
        const int32_t base = cut_offsets[first];
        for (int i = 0; i <= cnt; i++) o[i] = cut_offsets[first + i] - base;
        m->ext_states = g_states + first;
        m->ext_rhs = g_rhs + (size_t)first * cap;
        m->ext_rows = g_rows + (size_t)first * cap;
        const int r = relax_batch_impl(m, cnt, o.data(), type ? type + base : nullptr, var_index ? var_index + base : nullptr,
                                       value ? value + base : nullptr, check_cycles, out + first, nullptr, nullptr, 0, 1,
                                       want_rhs, want_rows, -1, compact);
        m->ext_states = nullptr; m->ext_rhs = nullptr; m->ext_rows = nullptr;
        if (r) return r;
        if (compact) {  // same layout on both sides: one copy per member
            if (rhs) memcpy(rhs + (size_t)first * cap, g_rhs + (size_t)first * cap, sizeof(double) * (size_t)cnt * cap);
            if (vibr) memcpy(vibr + (size_t)first * cap, g_rows + (size_t)first * cap, sizeof(int32_t) * (size_t)cnt * cap);
            return JSLP_OK;
        }
        for (int i = first; i < last && (rhs || vibr); i++) {  // caller-owned arrays: every member copies its own range
            const size_t H = (size_t)out[i].height;
            if (rhs) memcpy(rhs + (size_t)i * out_stride, g_rhs + (size_t)i * cap, sizeof(double) * H);
            if (vibr) memcpy(vibr + (size_t)i * out_stride, g_rows + (size_t)i * cap, sizeof(int32_t) * H);
        }
        return JSLP_OK;
    };
    for (int mi = 1; mi < M; mi++) p->workers[mi]->submit([job, mi] { return job(mi); });
    rc = job(0);
    rc = pool_join(p, rc);
    if (dbg_pool) {
        const double t_join = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_call).count();
        fprintf(stderr, "[jslp] pool call, %d nodes: joined at %.1f us; members (start -> end, us):", (int)n_nodes, t_join);
        for (int mi = 0; mi < M && mi < 64; mi++) fprintf(stderr, " %.1f -> %.1f", dbg_t[0][mi], dbg_t[1][mi]);
        fprintf(stderr, "\n");
    }
    hipSetDevice(e->device);
    return rc;
}

extern "C" int jslp_pool_relax_batch(jslp_pool* p, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                                     const int32_t* var_index, const double* value, int check_cycles,
                                     jslp_simplex_result* out, double* rhs, int32_t* var_index_by_row, int32_t out_stride) {
    return pool_relax(p, n_nodes, cut_offsets, type, var_index, value, check_cycles, out, rhs, var_index_by_row, out_stride,
                      rhs != nullptr, var_index_by_row != nullptr);
}

extern "C" int jslp_pool_relax_batch_pinned(jslp_pool* p, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                                            const int32_t* var_index, const double* value, int check_cycles,
                                            jslp_simplex_result* out, const double** rhs, const int32_t** var_index_by_row,
                                            int32_t* out_stride) {
    int rc = pool_relax(p, n_nodes, cut_offsets, type, var_index, value, check_cycles, out, nullptr, nullptr, 0, rhs != nullptr,
                        var_index_by_row != nullptr);
    if (rc) return rc;
    const size_t cap = (size_t)p->members[0]->cap_rows;
    if (rhs) *rhs = n_nodes > 0 ? reinterpret_cast<const double*>(p->h_out + (size_t)n_nodes * sizeof(DevState)) : nullptr;
    if (var_index_by_row)
        *var_index_by_row = n_nodes > 0 ? reinterpret_cast<const int32_t*>(p->h_out + (size_t)n_nodes * (sizeof(DevState) + cap * 8)) : nullptr;
    if (out_stride) *out_stride = (int32_t)cap;
    return JSLP_OK;
}

extern "C" int jslp_pool_set_watched_variables(jslp_pool* p, const int32_t* var_indexes, int32_t n) {
    if (!p) return fail(JSLP_ERR_ARG, "pool_set_watched_variables: null pool");
    jslp_engine* e = p->members[0];
    // the members learn their dimensions' worth of state with the root: make sure they have it before they are handed index lists
    if (e->uploaded && e->has_save && (!p->synced || p->synced_seq != e->root_seq)) {
        int rc = jslp_pool_sync_root(p);
        if (rc) return rc;
    }
    for (jslp_engine* m : p->members) {
        int rc = jslp_engine_set_watched_variables(m, var_indexes, n);
        if (rc) { hipSetDevice(e->device); return rc; }
    }
    hipSetDevice(e->device);
    return JSLP_OK;
}

extern "C" int jslp_pool_relax_batch_watched(jslp_pool* p, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                                             const int32_t* var_index, const double* value, int check_cycles,
                                             jslp_simplex_result* out, int32_t* watched_row, double* watched_value) {
    return pool_relax(p, n_nodes, cut_offsets, type, var_index, value, check_cycles, out, watched_value, watched_row, 0,
                      watched_value != nullptr, watched_row != nullptr, 1);
}

extern "C" int jslp_pool_relax_batch_watched_pinned(jslp_pool* p, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                                                    const int32_t* var_index, const double* value, int check_cycles,
                                                    jslp_simplex_result* out, const int32_t** watched_row, const double** watched_value) {
    int rc = pool_relax(p, n_nodes, cut_offsets, type, var_index, value, check_cycles, out, nullptr, nullptr, 0, watched_value != nullptr,
                        watched_row != nullptr, 1);
    if (rc) return rc;
    const size_t nw = p && !p->members.empty() ? (size_t)p->members[0]->n_watch : 0;
    if (watched_value) *watched_value = n_nodes > 0 ? reinterpret_cast<const double*>(p->h_out + (size_t)n_nodes * sizeof(DevState)) : nullptr;
    if (watched_row)
        *watched_row = n_nodes > 0 ? reinterpret_cast<const int32_t*>(p->h_out + (size_t)n_nodes * (sizeof(DevState) + nw * 8)) : nullptr;
    return JSLP_OK;
}

extern "C" int32_t jslp_pool_watched_count(const jslp_pool* p) {
    if (!p || p->members.empty()) return 0;
    const int32_t n = p->members[0]->n_watch;
    for (const jslp_engine* m : p->members)
        if (m->n_watch != n) return -1;
    return n;
}

extern "C" int jslp_pool_set_counting(jslp_pool* p, int enabled) {
    if (!p) return fail(JSLP_ERR_ARG, "pool_set_counting: null pool");
    for (jslp_engine* m : p->members) {
        int rc = jslp_engine_set_counting(m, enabled);
        if (rc) return rc;
    }
    hipSetDevice(p->members[0]->device);
    return JSLP_OK;
}

extern "C" int jslp_pool_get_counters(jslp_pool* p, jslp_work_counters* out) {
    if (!p || !out) return fail(JSLP_ERR_ARG, "pool_get_counters: null pointer");
    jslp_work_counters sum{};
    for (jslp_engine* m : p->members) {
        jslp_work_counters c;
        int rc = jslp_engine_get_counters(m, &c);
        if (rc) return rc;
        sum.relaxations += c.relaxations; sum.simplex_calls += c.simplex_calls; sum.pivots += c.pivots;
        sum.gated_cells += c.gated_cells; sum.gated_rows += c.gated_rows; sum.restored_rows += c.restored_rows;
        sum.cut_rows += c.cut_rows; sum.height_sum += c.height_sum;
        sum.resident_aborts += c.resident_aborts; sum.resident_handovers += c.resident_handovers; sum.resident_launches += c.resident_launches;
        sum.resident_refusals += c.resident_refusals; sum.node_queue_launches += c.node_queue_launches; sum.resident_fetch_retries += c.resident_fetch_retries;
    }
    hipSetDevice(p->members[0]->device);
    *out = sum;
    return JSLP_OK;
}

extern "C" int jslp_engine_dims(const jslp_engine* e, int32_t* height, int32_t* width, int32_t* n_var_indexes) {
    if (!e) return fail(JSLP_ERR_ARG, "dims: null engine");
    if (height) {
        if (hipSetDevice(e->device) != hipSuccess) return fail(JSLP_ERR_DEVICE, "dims: hipSetDevice failed");
        if (hipStreamSynchronize(e->stream) != hipSuccess) return fail(JSLP_ERR_DEVICE, "dims: stream synchronisation failed");
        DevState st;
        if (hipMemcpy(&st, e->s.st, sizeof st, hipMemcpyDeviceToHost) != hipSuccess)
            return fail(JSLP_ERR_DEVICE, "dims: state read-back failed");
        *height = e->uploaded ? st.H : e->H0;
    }
    if (width) *width = e->W;
    if (n_var_indexes) *n_var_indexes = e->n_idx;
    return JSLP_OK;
}

extern "C" int jslp_engine_download(jslp_engine* e, double* matrix, int32_t* var_index_by_row, int32_t* var_index_by_col,
                                    int32_t* row_by_var_index, int32_t* col_by_var_index) {
    if (!e || !e->uploaded) return fail(JSLP_ERR_STATE, "download before upload");
    HIPC(hipSetDevice(e->device));
    HIPC(hipStreamSynchronize(e->stream));
    DevState st;
    HIPC(hipMemcpy(&st, e->s.st, sizeof st, hipMemcpyDeviceToHost));
    const int H = st.H, W = e->W;
    if (matrix)
        HIPC(hipMemcpy2D(matrix, sizeof(double) * W, e->s.A, sizeof(double) * e->ld, sizeof(double) * W, H, hipMemcpyDeviceToHost));
    if (var_index_by_row) HIPC(hipMemcpy(var_index_by_row, e->s.vibr, sizeof(int32_t) * H, hipMemcpyDeviceToHost));
    if (var_index_by_col) HIPC(hipMemcpy(var_index_by_col, e->s.vibc, sizeof(int32_t) * W, hipMemcpyDeviceToHost));
    if (row_by_var_index) HIPC(hipMemcpy(row_by_var_index, e->s.rbv, sizeof(int32_t) * e->n_idx, hipMemcpyDeviceToHost));
    if (col_by_var_index) HIPC(hipMemcpy(col_by_var_index, e->s.cbv, sizeof(int32_t) * e->n_idx, hipMemcpyDeviceToHost));
    return JSLP_OK;
}

extern "C" int jslp_engine_pivot_trace(jslp_engine* e, int32_t* row_col, int64_t max_pairs, int64_t* n_pivots) {
    if (!e || !n_pivots) return fail(JSLP_ERR_ARG, "pivot_trace: null pointer");
    HIPC(hipSetDevice(e->device));
    HIPC(hipStreamSynchronize(e->stream));
    DevState st;
    HIPC(hipMemcpy(&st, e->s.st, sizeof st, hipMemcpyDeviceToHost));
    *n_pivots = st.trace_n;
    if (row_col && max_pairs > 0 && st.trace_n > e->s.trace_cap)
        return fail(JSLP_ERR_CAPACITY, "pivot_trace: more pivots since upload than the trace holds (2^20): the recorded prefix is not handed out as if it were complete");
    if (row_col && max_pairs > 0) {
        const long long n = std::min<long long>(std::min<long long>(st.trace_n, max_pairs), e->s.trace_cap);
        if (n > 0) HIPC(hipMemcpy(row_col, e->s.trace, sizeof(int2) * n, hipMemcpyDeviceToHost));
    }
    return JSLP_OK;
}

extern "C" const char* jslp_engine_last_path(const jslp_engine* e) { return e ? e->last_path : "none"; }

extern "C" int jslp_engine_set_timing(jslp_engine* e, int enabled) {
    if (!e) return fail(JSLP_ERR_ARG, "set_timing: null engine");
    e->timing = enabled ? 1 : 0;
    e->upd_ms = 0; e->upd_launches = 0; e->total_ms = 0;
    return JSLP_OK;
}

extern "C" int jslp_engine_get_timing(jslp_engine* e, double* update_kernel_ms, int64_t* update_kernel_launches,
                                      double* total_device_ms) {
    if (!e) return fail(JSLP_ERR_ARG, "get_timing: null engine");
    if (update_kernel_ms) *update_kernel_ms = e->upd_ms;
    if (update_kernel_launches) *update_kernel_launches = e->upd_launches;
    if (total_device_ms) *total_device_ms = e->total_ms;
    return JSLP_OK;
}
