/*
 * jslp_oracle.c -- TEST INFRASTRUCTURE ONLY.  Sequential C restatement of the reference's dense-tableau
 * hot path behind the C ABI of include/jslp_engine.h ("CPU fake device").
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (jslpsolver_amd/) never does and fails loudly when its HIP library is missing.
 *
 * Every function follows the reference file:line cited above it, statement by statement, in fp64 with the
 * same operation order (compile with -ffp-contract=off: JavaScript never fuses a*b+c).
 *
 * Parity is PINNED: tests/test_oracle_golden.py checks this file against the golden vectors produced by
 * the reference itself (oracle/_ref via tests/golden/gen_golden.js) -- every pivot (row, col) of all 47
 * reference fixtures and of the reference's synthetic generators, the sha256 of the final tableau, the
 * flags and every branch-and-bound relaxation outcome.
 */
#include "../include/jslp_engine.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct jslp_engine {
    int32_t width, height, height0, cap_rows, n_idx_cap;
    double precision;
    double* matrix;        /* cap_rows * width, row-major, stride width (tableau.ts:49-54)      */
    int32_t* vibr;         /* varIndexByRow  [cap_rows]                                          */
    int32_t* vibc;         /* varIndexByCol  [width]                                             */
    int32_t* rbv;          /* rowByVarIndex  [n_idx_cap]                                         */
    int32_t* cbv;          /* colByVarIndex  [n_idx_cap]                                         */
    uint8_t* unrestricted; /* unrestrictedVars[varIndex] === true  [n_idx_cap]                   */
    int32_t last_element_index;
    int uploaded;
    /* savedState (backup.ts:49-51) */
    int has_save;
    int32_t s_height, s_last_element_index;
    double* s_matrix;
    int32_t *s_vibr, *s_vibc, *s_rbv, *s_cbv;
    /* tableau scalars */
    int32_t feasible, bounded, unbounded_var_index;
    double evaluation;
    /* diagnostics */
    int32_t* trace;
    int64_t n_trace, cap_trace;
    int32_t* nz; /* nonZeroColumns scratch (simplex.ts:328) */
    /* optionalObjectives[o].reducedCosts (tableau.ts:71), sorted by priority; + their copy in savedState */
    int32_t n_opt;
    double* oo;
    double* s_oo;
    int32_t* defer; /* optionalCostsColumns scratch (simplex.ts:132-134) */
    uint8_t* is_int;       /* variablesPerIndex[v].isInteger  [n_idx_cap] */
    /* StateCheckpoint list (incremental-branch-and-cut.ts:31-44) */
    struct checkpoint* ck;
    int32_t n_ck;
    /* ABI extras mirrored so that host code written against the product library runs against this one */
    double* host_matrix;    /* jslp_engine_host_matrix */
    int32_t* watch;         /* jslp_engine_set_watched_variables */
    int32_t n_watch;
    int counting;
    jslp_work_counters wc;
    unsigned long long root_seq;
};

struct checkpoint {
    int live;
    int32_t height, last_element_index;
    double evaluation;
    double* matrix;
    int32_t *vibr, *vibc, *rbv, *cbv;
};

static void checkpoint_free(struct checkpoint* c) {
    free(c->matrix); free(c->vibr); free(c->vibc); free(c->rbv); free(c->cbv);
    memset(c, 0, sizeof *c);
}
static void checkpoints_clear(jslp_engine* e) {
    for (int32_t i = 0; i < e->n_ck; i++) checkpoint_free(&e->ck[i]);
    free(e->ck);
    e->ck = 0;
    e->n_ck = 0;
}

static __thread char g_err[256];
static int fail(int code, const char* msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}

const char* jslp_backend_name(void) { return "oracle-c"; }
void jslp_release_pooled_resources(void) {} /* nothing is pooled on the CPU */
const char* jslp_last_error(void) { return g_err; }
int jslp_device_count(void) { return 0; }

int jslp_engine_create(jslp_engine** out, int device, int32_t height, int32_t width, int32_t row_capacity,
                       double precision) {
    (void)device;
    if (!out || height < 1 || width < 1 || row_capacity < height) return fail(JSLP_ERR_ARG, "create: bad dims");
    jslp_engine* e = (jslp_engine*)calloc(1, sizeof *e);
    if (!e) return fail(JSLP_ERR_NOMEM, "create: oom");
    e->width = width;
    e->height = height;
    e->height0 = height;
    e->cap_rows = row_capacity;
    e->n_idx_cap = width + 2 * row_capacity + 2;
    e->precision = precision;
    size_t cells = (size_t)row_capacity * (size_t)width;
    e->matrix = (double*)calloc(cells, sizeof(double));
    e->s_matrix = (double*)calloc(cells, sizeof(double));
    e->vibr = (int32_t*)calloc(row_capacity, sizeof(int32_t));
    e->s_vibr = (int32_t*)calloc(row_capacity, sizeof(int32_t));
    e->vibc = (int32_t*)calloc(width, sizeof(int32_t));
    e->s_vibc = (int32_t*)calloc(width, sizeof(int32_t));
    e->rbv = (int32_t*)calloc(e->n_idx_cap, sizeof(int32_t));
    e->cbv = (int32_t*)calloc(e->n_idx_cap, sizeof(int32_t));
    e->s_rbv = (int32_t*)calloc(e->n_idx_cap, sizeof(int32_t));
    e->s_cbv = (int32_t*)calloc(e->n_idx_cap, sizeof(int32_t));
    e->unrestricted = (uint8_t*)calloc(e->n_idx_cap, 1);
    e->nz = (int32_t*)calloc(width, sizeof(int32_t));
    if (!e->matrix || !e->s_matrix || !e->vibr || !e->s_vibr || !e->vibc || !e->s_vibc || !e->rbv || !e->cbv ||
        !e->s_rbv || !e->s_cbv || !e->unrestricted || !e->nz) {
        jslp_engine_destroy(e);
        return fail(JSLP_ERR_NOMEM, "create: oom");
    }
    e->feasible = 1;
    e->bounded = 1;
    e->unbounded_var_index = -1;
    *out = e;
    return JSLP_OK;
}

void jslp_engine_destroy(jslp_engine* e) {
    if (!e) return;
    free(e->matrix); free(e->s_matrix); free(e->vibr); free(e->s_vibr); free(e->vibc); free(e->s_vibc);
    free(e->rbv); free(e->cbv); free(e->s_rbv); free(e->s_cbv); free(e->unrestricted); free(e->nz);
    free(e->trace);
    free(e->oo); free(e->s_oo); free(e->defer);
    checkpoints_clear(e);
    free(e->is_int);
    free(e->host_matrix);
    free(e->watch);
    free(e);
}

/* Tableau.initialize + the index-map part of _resetMatrix (tableau.ts:292-317, 341-357) */
int jslp_engine_upload(jslp_engine* e, const double* matrix, const int32_t* var_index_by_row,
                       const int32_t* var_index_by_col, const int32_t* unrestricted_var_indexes,
                       int32_t n_unrestricted) {
    if (!e || !matrix || !var_index_by_row || !var_index_by_col) return fail(JSLP_ERR_ARG, "upload: null");
    e->height = e->height0; /* cuts may have grown the tableau: an upload starts from the created height */
    const int32_t H = e->height, W = e->width;
    memcpy(e->matrix, matrix, (size_t)H * W * sizeof(double));
    for (int32_t i = 0; i < e->n_idx_cap; i++) { e->rbv[i] = -1; e->cbv[i] = -1; e->unrestricted[i] = 0; }
    e->vibr[0] = -1;
    e->vibc[0] = -1;
    for (int32_t r = 1; r < H; r++) {
        int32_t v = var_index_by_row[r];
        if (v < 0 || v >= e->n_idx_cap) return fail(JSLP_ERR_ARG, "upload: row var index out of range");
        e->vibr[r] = v;
        e->rbv[v] = r;
    }
    for (int32_t c = 1; c < W; c++) {
        int32_t v = var_index_by_col[c];
        if (v < 0 || v >= e->n_idx_cap) return fail(JSLP_ERR_ARG, "upload: col var index out of range");
        e->vibc[c] = v;
        e->cbv[v] = c;
    }
    for (int32_t i = 0; i < n_unrestricted; i++) {
        int32_t v = unrestricted_var_indexes[i];
        if (v < 0 || v >= e->n_idx_cap) return fail(JSLP_ERR_ARG, "upload: unrestricted index out of range");
        e->unrestricted[v] = 1;
    }
    e->last_element_index = W + H - 2; /* tableau.ts:312-316 */
    e->has_save = 0;
    checkpoints_clear(e);
    if (e->is_int) memset(e->is_int, 0, (size_t)e->n_idx_cap);
    e->feasible = 1;
    e->bounded = 1;
    e->evaluation = 0;
    e->unbounded_var_index = -1;
    e->n_trace = 0;
    e->uploaded = 1;
    e->n_opt = 0;
    e->root_seq += 1;
    return JSLP_OK;
}

int jslp_engine_set_optional_objectives(jslp_engine* e, int32_t n, const double* rows) {
    if (!e || !e->uploaded) return fail(JSLP_ERR_STATE, "set_optional_objectives before upload");
    if (n < 0 || (n > 0 && !rows)) return fail(JSLP_ERR_ARG, "set_optional_objectives: bad arguments");
    free(e->oo); free(e->s_oo); free(e->defer);
    e->oo = (double*)calloc((size_t)(n > 0 ? n : 1) * e->width, sizeof(double));
    e->s_oo = (double*)calloc((size_t)(n > 0 ? n : 1) * e->width, sizeof(double));
    e->defer = (int32_t*)calloc((size_t)e->width * 2, sizeof(int32_t));
    if (!e->oo || !e->s_oo || !e->defer) return fail(JSLP_ERR_NOMEM, "set_optional_objectives: oom");
    if (n > 0) memcpy(e->oo, rows, (size_t)n * e->width * sizeof(double));
    e->n_opt = n;
    return JSLP_OK;
}

int jslp_engine_get_optional_objectives(jslp_engine* e, double* rows, int32_t* n_out) {
    if (!e || !e->uploaded) return fail(JSLP_ERR_STATE, "get_optional_objectives before upload");
    if (n_out) *n_out = e->n_opt;
    if (rows && e->n_opt > 0) memcpy(rows, e->oo, (size_t)e->n_opt * e->width * sizeof(double));
    return JSLP_OK;
}

static void trace_push(jslp_engine* e, int32_t r, int32_t c) {
    if (e->n_trace + 1 > e->cap_trace) {
        int64_t nc = e->cap_trace ? e->cap_trace * 2 : 1024;
        int32_t* p = (int32_t*)realloc(e->trace, (size_t)nc * 2 * sizeof(int32_t));
        if (!p) return;
        e->trace = p;
        e->cap_trace = nc;
    }
    e->trace[2 * e->n_trace] = r;
    e->trace[2 * e->n_trace + 1] = c;
    e->n_trace++;
}

/* the reference's zero test `!(v >= -1e-16 && v <= 1e-16)` (simplex.ts:356,372,375,379): NaN counts as non-zero */
static inline int nonzero16(double v) { return !(v >= -1e-16 && v <= 1e-16); }

/* pivot (simplex.ts:330-413) */
static void pivot(jslp_engine* e, int32_t pr, int32_t pc) {
    double* m = e->matrix;
    const int32_t width = e->width, height = e->height;
    const size_t pro = (size_t)pr * width;
    const double quotient = m[pro + pc]; /* :335 */

    const int32_t leaving = e->vibr[pr], entering = e->vibc[pc]; /* :339-340 */
    e->vibr[pr] = entering;
    e->vibc[pc] = leaving;
    e->rbv[entering] = pr;
    e->rbv[leaving] = -1;
    e->cbv[entering] = -1;
    e->cbv[leaving] = pc;

    int32_t nnz = 0; /* :352-363 */
    for (int32_t c = 0; c < width; c++) {
        const double val = m[pro + c];
        if (nonzero16(val)) {
            m[pro + c] = val / quotient;
            e->nz[nnz++] = c;
        } else {
            m[pro + c] = 0;
        }
    }
    m[pro + pc] = 1 / quotient; /* :364 */

    int64_t gated_rows = 0; /* work counters only */
    for (int32_t r = 0; r < height; r++) { /* :367-392 */
        if (r == pr) continue;
        const size_t ro = (size_t)r * width;
        const double pcv = m[ro + pc];
        if (nonzero16(pcv)) {
            gated_rows += 1;
            const double coefficient = pcv;
            for (int32_t i = 0; i < nnz; i++) {
                const int32_t c = e->nz[i];
                const double v0 = m[pro + c];
                if (nonzero16(v0)) {
                    m[ro + c] = m[ro + c] - coefficient * v0; /* two roundings */
                } else if (v0 != 0) {
                    m[pro + c] = 0;
                }
            }
            m[ro + pc] = -coefficient / quotient;
        }
        /* the reference's inner `else if (coefficient !== 0) matrix[...] = 0` (:388-390) is unreachable: it sits
           inside `if (!(pivotColVal tiny))` with coefficient === pivotColVal, so tiny entries stay untouched */
    }
    if (e->counting) { /* jslp_work_counters: gated rows x live columns of the final pivot row (plus c*) */
        int64_t cols = 0;
        for (int32_t c = 0; c < width; c++) cols += (nonzero16(m[pro + c]) || c == pc) ? 1 : 0;
        e->wc.gated_rows += gated_rows;
        e->wc.gated_cells += gated_rows * cols;
    }
    /* optional objectives (:394-412): exact `!== 0` tests instead of the 1e-16 band */
    for (int32_t o = 0; o < e->n_opt; o++) {
        double* rc = e->oo + (size_t)o * width;
        const double coefficient = rc[pc];
        if (coefficient != 0) {
            for (int32_t i = 0; i < nnz; i++) {
                const int32_t c = e->nz[i];
                const double v0 = m[pro + c];
                if (v0 != 0) rc[c] = rc[c] - coefficient * v0;
            }
            rc[pc] = -coefficient / quotient;
        }
    }
    trace_push(e, pr, pc);
}

/* checkForCycles (simplex.ts:415-440): returns 1 and fills start/len when a repeated block is found */
static int check_for_cycles(const int32_t* h /* pairs */, int64_t n, int32_t* start, int32_t* len) {
    for (int64_t e1 = 0; e1 < n - 1; e1++) {
        for (int64_t e2 = e1 + 1; e2 < n; e2++) {
            if (h[2 * e1] == h[2 * e2] && h[2 * e1 + 1] == h[2 * e2 + 1]) {
                if (e2 - e1 > n - e2) break;
                int found = 1;
                for (int64_t i = 1; i < e2 - e1; i++) {
                    if (h[2 * (e1 + i)] != h[2 * (e2 + i)] || h[2 * (e1 + i) + 1] != h[2 * (e2 + i) + 1]) {
                        found = 0;
                        break;
                    }
                }
                if (found) {
                    *start = (int32_t)e1;
                    *len = (int32_t)(e2 - e1);
                    return 1;
                }
            }
        }
    }
    return 0;
}

typedef struct { int32_t* h; int64_t n, cap; } hist_t;
static void hist_push(hist_t* hs, int32_t a, int32_t b) {
    if (hs->n + 1 > hs->cap) {
        hs->cap = hs->cap ? hs->cap * 2 : 256;
        hs->h = (int32_t*)realloc(hs->h, (size_t)hs->cap * 2 * sizeof(int32_t));
    }
    hs->h[2 * hs->n] = a;
    hs->h[2 * hs->n + 1] = b;
    hs->n++;
}

/* phase1 (simplex.ts:25-98) */
static int32_t phase1(jslp_engine* e, int check_cycles, jslp_simplex_result* res) {
    const double* m = e->matrix;
    const int32_t width = e->width, last_col = e->width - 1, last_row = e->height - 1;
    const double precision = e->precision;
    hist_t hs = {0, 0, 0};
    int32_t iterations = 0;
    for (;;) {
        int32_t leaving_row = 0; /* :39-49 */
        double rhs_value = -precision;
        for (int32_t r = 1; r <= last_row; r++) {
            const double value = m[(size_t)r * width];
            if (value < rhs_value) {
                rhs_value = value;
                leaving_row = r;
            }
        }
        if (leaving_row == 0) { /* :51-54 */
            e->feasible = 1;
            break;
        }
        int32_t entering_col = 0; /* :56-71 */
        double max_quotient = -INFINITY;
        const size_t lro = (size_t)leaving_row * width;
        for (int32_t c = 1; c <= last_col; c++) {
            const double coefficient = m[lro + c];
            const int unrestricted = e->unrestricted[e->vibc[c]];
            if (unrestricted || coefficient < -precision) {
                const double quotient = -m[c] / coefficient;
                if (max_quotient < quotient) {
                    max_quotient = quotient;
                    entering_col = c;
                }
            }
        }
        if (entering_col == 0) { /* :73-76 */
            e->feasible = 0;
            break;
        }
        if (check_cycles) { /* :78-93 */
            hist_push(&hs, e->vibr[leaving_row], e->vibc[entering_col]);
            int32_t s, l;
            if (check_for_cycles(hs.h, hs.n, &s, &l)) {
                res->cycle_phase = 1;
                res->cycle_start = s;
                res->cycle_length = l;
                e->feasible = 0;
                break;
            }
        }
        pivot(e, leaving_row, entering_col);
        m = e->matrix;
        iterations++;
    }
    free(hs.h);
    return iterations;
}

/* Math.round: nearest integer, ties toward +Infinity */
static double js_round(double x) {
    if (!isfinite(x)) return x;
    double f = floor(x);
    return (x - f >= 0.5) ? f + 1.0 : f;
}

/* setEvaluation (tableau.ts:420-430) */
static double rounded_evaluation(const jslp_engine* e) {
    const double rc = js_round(1 / e->precision);
    return js_round((2.220446049250313e-16 + e->matrix[0]) * rc) / rc;
}

/* phase2 (simplex.ts:100-325) */
static int32_t phase2(jslp_engine* e, int check_cycles, jslp_simplex_result* res) {
    const double* m = e->matrix;
    const int32_t width = e->width, last_col = e->width - 1, last_row = e->height - 1;
    const double precision = e->precision;
    hist_t hs = {0, 0, 0};
    int32_t iterations = 0;

    const int32_t n_columns = last_col; /* :118-127 */
    int32_t batch = (int32_t)floor(sqrt((double)n_columns));
    if (batch < 50) batch = 50;
    if (batch > 500) batch = 500;
    const int use_partial = n_columns > batch * 2;
    int32_t pricing_batch_start = 1; /* tableau.ts:91; provably always 1 at loop entry */

    for (;;) {
        int32_t entering_col = 0; /* :136-219 */
        double entering_value = precision;
        int is_rc_negative = 0;
        const int32_t n_opt = e->n_opt;
        int32_t n_defer = 0; /* optionalCostsColumns (:132-134) */
        if (use_partial) {
            const int32_t start_batch = pricing_batch_start;
            int32_t scanned = 0;
            const int32_t total = (n_columns + batch - 1) / batch;
            while (entering_col == 0 && scanned < total) {
                const int32_t bs = pricing_batch_start;
                int32_t be = bs + batch - 1;
                if (be > last_col) be = last_col;
                for (int32_t c = bs; c <= be; c++) {
                    const double rc = m[c];
                    const int unrestricted = e->unrestricted[e->vibc[c]];
                    if (n_opt > 0 && -precision < rc && rc < precision) { /* :155-162 */
                        e->defer[n_defer++] = c;
                        continue;
                    }
                    if (unrestricted && rc < 0) {
                        if (-rc > entering_value) {
                            entering_value = -rc;
                            entering_col = c;
                            is_rc_negative = 1;
                        }
                        continue;
                    }
                    if (rc > entering_value) {
                        entering_value = rc;
                        entering_col = c;
                        is_rc_negative = 0;
                    }
                }
                pricing_batch_start = be >= last_col ? 1 : be + 1;
                scanned++;
            }
            if (entering_col != 0) pricing_batch_start = start_batch;
        } else {
            for (int32_t c = 1; c <= last_col; c++) {
                const double rc = m[c];
                const int unrestricted = e->unrestricted[e->vibc[c]];
                if (n_opt > 0 && -precision < rc && rc < precision) { /* :195-202 */
                    e->defer[n_defer++] = c;
                    continue;
                }
                if (unrestricted && rc < 0) {
                    if (-rc > entering_value) {
                        entering_value = -rc;
                        entering_col = c;
                        is_rc_negative = 1;
                    }
                    continue;
                }
                if (rc > entering_value) {
                    entering_value = rc;
                    entering_col = c;
                    is_rc_negative = 0;
                }
            }
        }
        if (n_opt > 0) { /* :221-263: break ties on the priority-ordered secondary objectives */
            int32_t o = 0;
            int32_t* cur = e->defer;
            int32_t* nxt = e->defer + e->width;
            while (entering_col == 0 && n_defer > 0 && o < n_opt) {
                const double* rcs = e->oo + (size_t)o * width;
                int32_t n_next = 0;
                entering_value = precision;
                for (int32_t i = 0; i < n_defer; i++) {
                    const int32_t c = cur[i];
                    const double rc = rcs[c];
                    const int unrestricted = e->unrestricted[e->vibc[c]];
                    if (-precision < rc && rc < precision) {
                        nxt[n_next++] = c;
                        continue;
                    }
                    if (unrestricted && rc < 0) {
                        if (-rc > entering_value) {
                            entering_value = -rc;
                            entering_col = c;
                            is_rc_negative = 1;
                        }
                        continue;
                    }
                    if (rc > entering_value) {
                        entering_value = rc;
                        entering_col = c;
                        is_rc_negative = 0;
                    }
                }
                int32_t* t = cur; cur = nxt; nxt = t;
                n_defer = n_next;
                o += 1;
            }
        }
        if (entering_col == 0) { /* :265-269 */
            e->evaluation = rounded_evaluation(e);
            res->optimal = 1;
            break;
        }
        int32_t leaving_row = 0; /* :271-296 */
        double min_quotient = INFINITY;
        for (int32_t r = 1; r <= last_row; r++) {
            const size_t ro = (size_t)r * width;
            const double rhs = m[ro];
            const double col = m[ro + entering_col];
            if (-precision < col && col < precision) continue;
            if (col > 0 && precision > rhs && rhs > -precision) {
                min_quotient = 0;
                leaving_row = r;
                break;
            }
            const double quotient = is_rc_negative ? -rhs / col : rhs / col;
            if (quotient > precision && min_quotient > quotient) {
                min_quotient = quotient;
                leaving_row = r;
            }
        }
        if (min_quotient == INFINITY) { /* :298-303 */
            e->evaluation = -INFINITY;
            e->bounded = 0;
            e->unbounded_var_index = e->vibc[entering_col];
            break;
        }
        if (check_cycles) { /* :305-320 */
            hist_push(&hs, e->vibr[leaving_row], e->vibc[entering_col]);
            int32_t s, l;
            if (check_for_cycles(hs.h, hs.n, &s, &l)) {
                res->cycle_phase = 2;
                res->cycle_start = s;
                res->cycle_length = l;
                e->feasible = 0;
                break;
            }
        }
        pivot(e, leaving_row, entering_col);
        m = e->matrix;
        iterations++;
    }
    free(hs.h);
    return iterations;
}

/* simplex (simplex.ts:14-23) */
int jslp_engine_simplex(jslp_engine* e, int check_cycles, jslp_simplex_result* out) {
    if (!e || !out) return fail(JSLP_ERR_ARG, "simplex: null");
    if (!e->uploaded) return fail(JSLP_ERR_STATE, "simplex before upload");
    memset(out, 0, sizeof *out);
    out->pivots_phase2 = -1;
    e->bounded = 1;
    out->pivots_phase1 = phase1(e, check_cycles, out);
    if (e->feasible) out->pivots_phase2 = phase2(e, check_cycles, out);
    out->feasible = e->feasible;
    out->bounded = e->bounded;
    out->unbounded_var_index = e->bounded ? -1 : e->unbounded_var_index;
    out->height = e->height;
    out->obj_cell = e->matrix[0];
    out->evaluation = e->evaluation;
    if (e->counting) {
        e->wc.simplex_calls += 1;
        e->wc.pivots += out->pivots_phase1 + (out->pivots_phase2 > 0 ? out->pivots_phase2 : 0);
        e->wc.height_sum += e->height;
    }
    return JSLP_OK;
}

int jslp_engine_pivot(jslp_engine* e, int32_t row, int32_t col) {
    if (!e || !e->uploaded) return fail(JSLP_ERR_STATE, "pivot before upload");
    if (row < 0 || row >= e->height || col < 0 || col >= e->width) return fail(JSLP_ERR_ARG, "pivot: out of range");
    pivot(e, row, col);
    return JSLP_OK;
}

/* save = copy (backup.ts:13-51) */
int jslp_engine_save(jslp_engine* e) {
    if (!e || !e->uploaded) return fail(JSLP_ERR_STATE, "save before upload");
    memcpy(e->s_matrix, e->matrix, (size_t)e->height * e->width * sizeof(double));
    memcpy(e->s_vibr, e->vibr, (size_t)e->height * sizeof(int32_t));
    memcpy(e->s_vibc, e->vibc, (size_t)e->width * sizeof(int32_t));
    memcpy(e->s_rbv, e->rbv, (size_t)e->n_idx_cap * sizeof(int32_t));
    memcpy(e->s_cbv, e->cbv, (size_t)e->n_idx_cap * sizeof(int32_t));
    if (e->n_opt > 0) memcpy(e->s_oo, e->oo, (size_t)e->n_opt * e->width * sizeof(double)); /* backup.ts:37-43 */
    e->s_height = e->height;
    e->s_last_element_index = e->last_element_index;
    e->has_save = 1;
    e->root_seq += 1;
    return JSLP_OK;
}

/* restore (backup.ts:53-105): flags / evaluation are NOT part of the snapshot */
int jslp_engine_restore(jslp_engine* e) {
    if (!e || !e->uploaded) return fail(JSLP_ERR_STATE, "restore before upload");
    if (!e->has_save) return JSLP_OK; /* :54-56 */
    e->height = e->s_height;
    e->last_element_index = e->s_last_element_index;
    memcpy(e->matrix, e->s_matrix, (size_t)e->height * e->width * sizeof(double));
    memcpy(e->vibr, e->s_vibr, (size_t)e->height * sizeof(int32_t));
    memcpy(e->vibc, e->s_vibc, (size_t)e->width * sizeof(int32_t));
    /* the reference restores rowByVarIndex/colByVarIndex for v < save.nVars only (:87-92); cut slacks above
       that keep stale entries there, which nothing reads before addCutConstraints rewrites them */
    memcpy(e->rbv, e->s_rbv, (size_t)e->n_idx_cap * sizeof(int32_t));
    memcpy(e->cbv, e->s_cbv, (size_t)e->n_idx_cap * sizeof(int32_t));
    if (e->n_opt > 0) memcpy(e->oo, e->s_oo, (size_t)e->n_opt * e->width * sizeof(double)); /* backup.ts:94-104 */
    if (e->counting) e->wc.restored_rows += e->height; /* the reference copies every row; the product only the changed ones */
    return JSLP_OK;
}

/* addCutConstraints (cutting-strategies.ts:16-72) */
int jslp_engine_add_cuts(jslp_engine* e, int32_t n, const int8_t* type, const int32_t* var_index,
                         const double* value) {
    if (!e || !e->uploaded) return fail(JSLP_ERR_STATE, "add_cuts before upload");
    if (n < 0 || (n > 0 && (!type || !var_index || !value))) return fail(JSLP_ERR_ARG, "add_cuts: null");
    if (e->height + n > e->cap_rows) return fail(JSLP_ERR_CAPACITY, "add_cuts: row capacity exceeded");
    double* m = e->matrix;
    const int32_t width = e->width, height = e->height, last_col = e->width - 1;
    for (int32_t h = 0; h < n; h++) {
        const int32_t cut_row = height + h;
        const size_t cro = (size_t)cut_row * width;
        const double sign = type[h] == JSLP_CUT_MIN ? -1 : 1; /* :41 */
        const int32_t vi = var_index[h];
        if (vi < 0 || vi >= e->n_idx_cap) return fail(JSLP_ERR_ARG, "add_cuts: var index out of range");
        const int32_t var_row = e->rbv[vi];
        if (var_row == -1) { /* :46-53 */
            if (e->cbv[vi] < 0) return fail(JSLP_ERR_ARG, "add_cuts: variable neither basic nor non-basic");
            m[cro] = sign * value[h];
            for (int32_t c = 1; c <= last_col; c++) m[cro + c] = 0;
            m[cro + e->cbv[vi]] = sign;
        } else { /* :54-62 */
            const size_t vro = (size_t)var_row * width;
            const double var_value = m[vro];
            m[cro] = sign * (value[h] - var_value);
            for (int32_t c = 1; c <= last_col; c++) m[cro + c] = -sign * m[vro + c];
        }
        const int32_t slack = e->last_element_index++; /* getNewElementIndex, tableau.ts:393-401 */
        if (slack >= e->n_idx_cap) return fail(JSLP_ERR_CAPACITY, "add_cuts: element index capacity exceeded");
        e->vibr[cut_row] = slack;
        e->rbv[slack] = cut_row;
        e->cbv[slack] = -1;
    }
    e->height = height + n;
    if (e->counting) e->wc.cut_rows += n;
    return JSLP_OK;
}

int jslp_engine_read_rhs(jslp_engine* e, double* rhs, int32_t* var_index_by_row) {
    if (!e || !e->uploaded) return fail(JSLP_ERR_STATE, "read_rhs before upload");
    for (int32_t r = 0; r < e->height; r++) {
        if (rhs) rhs[r] = e->matrix[(size_t)r * e->width];
        if (var_index_by_row) var_index_by_row[r] = e->vibr[r];
    }
    return JSLP_OK;
}

/* applyCuts (branch-and-cut.ts:33-37) */
int jslp_engine_relax(jslp_engine* e, int32_t n_cuts, const int8_t* type, const int32_t* var_index,
                      const double* value, int check_cycles, jslp_simplex_result* out, double* rhs,
                      int32_t* var_index_by_row) {
    int rc = jslp_engine_restore(e);
    if (rc) return rc;
    rc = jslp_engine_add_cuts(e, n_cuts, type, var_index, value);
    if (rc) return rc;
    rc = jslp_engine_simplex(e, check_cycles, out);
    if (rc) return rc;
    if (e->counting) e->wc.relaxations += 1;
    return jslp_engine_read_rhs(e, rhs, var_index_by_row);
}

int jslp_engine_relax_batch(jslp_engine* e, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                            const int32_t* var_index, const double* value, int check_cycles,
                            jslp_simplex_result* out, double* rhs, int32_t* var_index_by_row,
                            int32_t out_stride) {
    if (!e || n_nodes < 0 || !cut_offsets || !out) return fail(JSLP_ERR_ARG, "relax_batch: null");
    if ((rhs || var_index_by_row) && out_stride < e->cap_rows) return fail(JSLP_ERR_ARG, "relax_batch: out_stride < row capacity");
    /* each node starts from the saved root, but flags carry over exactly as in a sequential B&B only through
       `evaluation` of non-optimal nodes; nodes here are independent, so reset it per node */
    for (int32_t i = 0; i < n_nodes; i++) {
        const int32_t a = cut_offsets[i], n = cut_offsets[i + 1] - a;
        int rc = jslp_engine_relax(e, n, type ? type + a : 0, var_index ? var_index + a : 0, value ? value + a : 0,
                                   check_cycles, &out[i], rhs ? rhs + (size_t)i * out_stride : 0,
                                   var_index_by_row ? var_index_by_row + (size_t)i * out_stride : 0);
        if (rc) return rc;
    }
    return JSLP_OK;
}

/* zero-copy variant of the ABI: here simply backed by heap buffers owned by the engine */
int jslp_engine_relax_batch_pinned(jslp_engine* e, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                                   const int32_t* var_index, const double* value, int check_cycles,
                                   jslp_simplex_result* out, const double** rhs, const int32_t** var_index_by_row,
                                   int32_t* out_stride) {
    if (!e) return fail(JSLP_ERR_ARG, "relax_batch_pinned: null");
    static __thread double* b_rhs = 0;
    static __thread int32_t* b_rows = 0;
    static __thread size_t b_cap = 0;
    const size_t need = (size_t)(n_nodes > 0 ? n_nodes : 1) * e->cap_rows;
    if (need > b_cap) {
        b_rhs = (double*)realloc(b_rhs, need * sizeof(double));
        b_rows = (int32_t*)realloc(b_rows, need * sizeof(int32_t));
        b_cap = need;
    }
    int rc = jslp_engine_relax_batch(e, n_nodes, cut_offsets, type, var_index, value, check_cycles, out, b_rhs, b_rows,
                                     e->cap_rows);
    if (rc) return rc;
    if (rhs) *rhs = b_rhs;
    if (var_index_by_row) *var_index_by_row = b_rows;
    if (out_stride) *out_stride = e->cap_rows;
    return JSLP_OK;
}

int jslp_engine_set_integer_variables(jslp_engine* e, const int32_t* var_indexes, int32_t n) {
    if (!e || !e->uploaded) return fail(JSLP_ERR_STATE, "set_integer_variables before upload");
    if (n < 0 || (n > 0 && !var_indexes)) return fail(JSLP_ERR_ARG, "set_integer_variables: bad arguments");
    if (!e->is_int) e->is_int = (uint8_t*)calloc((size_t)e->n_idx_cap, 1);
    if (!e->is_int) return fail(JSLP_ERR_NOMEM, "set_integer_variables: out of memory");
    memset(e->is_int, 0, (size_t)e->n_idx_cap);
    for (int32_t i = 0; i < n; i++) {
        if (var_indexes[i] < 0 || var_indexes[i] >= e->n_idx_cap) return fail(JSLP_ERR_ARG, "set_integer_variables: index out of range");
        e->is_int[var_indexes[i]] = 1;
    }
    return JSLP_OK;
}

static int is_integer_var(const jslp_engine* e, int32_t v) { return v >= 0 && e->is_int && e->is_int[v]; }

/* Math.max(0, x) / Math.min(0, x) with JavaScript's treatment of NaN and signed zeros */
static double js_max0(double x) { return x != x ? x : (x > 0 ? x : 0.0); }
static double js_min0(double x) { return x != x ? x : ((x < 0 || (x == 0 && signbit(x))) ? x : 0.0); }

/* addLowerBoundMIRCut (cutting-strategies.ts:74-135); 1 = a row was appended */
static int add_lower_bound_mir_cut(jslp_engine* e, int32_t row, int* err) {
    if (row == 0) return 0; /* costRowIndex */
    const int32_t width = e->width;
    double* m = e->matrix;
    const size_t src = (size_t)row * width;
    if (!is_integer_var(e, e->vibr[row])) return 0; /* :82-85 */
    const double rhs = m[src];
    const double f = rhs - floor(rhs);
    if (f < e->precision || f > 1 - e->precision) return 0; /* :88-90 */
    const int32_t height = e->height;
    if (height + 1 > e->cap_rows) { *err = JSLP_ERR_CAPACITY; return 0; }
    const size_t dst = (size_t)height * width;
    e->height += 1;
    const int32_t slack = e->last_element_index++; /* getNewElementIndex */
    if (slack >= e->n_idx_cap) { *err = JSLP_ERR_CAPACITY; return 0; }
    e->vibr[height] = slack;
    e->rbv[slack] = height;
    e->cbv[slack] = -1;
    m[dst] = floor(rhs); /* :112 */
    for (int32_t c = 1; c < width; c++) { /* :114-126 */
        const double a = m[src + c];
        if (is_integer_var(e, e->vibc[c])) {
            const double fl = floor(a);
            m[dst + c] = fl + js_max0(a - fl - f) / (1 - f);
        } else {
            m[dst + c] = js_min0(a / (1 - f));
        }
    }
    for (int32_t c = 0; c < width; c++) m[dst + c] -= m[src + c]; /* :128-130 */
    return 1;
}

/* applyMIRCuts (cutting-strategies.ts:199-212) */
int jslp_engine_apply_mir_cuts(jslp_engine* e, int32_t* n_added) {
    if (!e || !e->uploaded) return fail(JSLP_ERR_STATE, "apply_mir_cuts before upload");
    const int32_t height = e->height;
    int32_t added = 0;
    int err = 0;
    for (int32_t r = 1; r < height && added < 10; r++) {
        if (add_lower_bound_mir_cut(e, r, &err)) added++;
        if (err) return fail(err, "apply_mir_cuts: row capacity exceeded");
    }
    if (n_added) *n_added = added;
    return JSLP_OK;
}

int jslp_engine_mir_round(jslp_engine* e, int check_cycles, int32_t* n_added, jslp_simplex_result* out, double* rhs,
                          int32_t* var_index_by_row) {
    int rc = jslp_engine_apply_mir_cuts(e, n_added);
    if (rc) return rc;
    rc = jslp_engine_simplex(e, check_cycles, out);
    if (rc) return rc;
    return jslp_engine_read_rhs(e, rhs, var_index_by_row);
}

/* the fp32 experiment has no reference semantics to restate */
int jslp_engine_simplex_f32(jslp_engine* e, double precision, int check_cycles, jslp_simplex_result* out, double* rhs,
                            int32_t* var_index_by_row, double* device_ms) {
    (void)e; (void)precision; (void)check_cycles; (void)out; (void)rhs; (void)var_index_by_row; (void)device_ms;
    return fail(JSLP_ERR_UNSUPPORTED, "simplex_f32: not part of the reference (HIP engine only)");
}

/* createCheckpoint (incremental-branch-and-cut.ts:55-70) */
int jslp_engine_checkpoint_create(jslp_engine* e, int32_t* id_out) {
    if (!e || !id_out) return fail(JSLP_ERR_ARG, "checkpoint_create: null");
    if (!e->uploaded) return fail(JSLP_ERR_STATE, "checkpoint before upload");
    int32_t id = -1;
    for (int32_t i = 0; i < e->n_ck; i++)
        if (!e->ck[i].live) { id = i; break; }
    if (id < 0) {
        struct checkpoint* g = (struct checkpoint*)realloc(e->ck, (size_t)(e->n_ck + 1) * sizeof *g);
        if (!g) return fail(JSLP_ERR_NOMEM, "checkpoint_create: out of memory");
        e->ck = g;
        id = e->n_ck++;
        memset(&e->ck[id], 0, sizeof e->ck[id]);
    }
    struct checkpoint* c = &e->ck[id];
    const size_t cells = (size_t)e->height * e->width;
    c->matrix = (double*)malloc(cells * sizeof(double));
    c->vibr = (int32_t*)malloc((size_t)e->height * sizeof(int32_t));
    c->vibc = (int32_t*)malloc((size_t)e->width * sizeof(int32_t));
    c->rbv = (int32_t*)malloc((size_t)e->n_idx_cap * sizeof(int32_t));
    c->cbv = (int32_t*)malloc((size_t)e->n_idx_cap * sizeof(int32_t));
    if (!c->matrix || !c->vibr || !c->vibc || !c->rbv || !c->cbv) {
        checkpoint_free(c);
        return fail(JSLP_ERR_NOMEM, "checkpoint_create: out of memory");
    }
    memcpy(c->matrix, e->matrix, cells * sizeof(double));
    memcpy(c->vibr, e->vibr, (size_t)e->height * sizeof(int32_t));
    memcpy(c->vibc, e->vibc, (size_t)e->width * sizeof(int32_t));
    memcpy(c->rbv, e->rbv, (size_t)e->n_idx_cap * sizeof(int32_t));
    memcpy(c->cbv, e->cbv, (size_t)e->n_idx_cap * sizeof(int32_t));
    c->height = e->height;
    c->last_element_index = e->last_element_index;
    c->evaluation = e->evaluation;
    c->live = 1;
    *id_out = id;
    return JSLP_OK;
}

/* restoreCheckpoint (incremental-branch-and-cut.ts:72-107): optional objectives and the saved root stay as they are */
int jslp_engine_checkpoint_restore(jslp_engine* e, int32_t id) {
    if (!e || !e->uploaded) return fail(JSLP_ERR_STATE, "checkpoint_restore before upload");
    if (id < 0 || id >= e->n_ck || !e->ck[id].live) return fail(JSLP_ERR_ARG, "checkpoint_restore: no such checkpoint");
    const struct checkpoint* c = &e->ck[id];
    e->height = c->height;
    e->last_element_index = c->last_element_index;
    e->evaluation = c->evaluation;
    memcpy(e->matrix, c->matrix, (size_t)c->height * e->width * sizeof(double));
    memcpy(e->vibr, c->vibr, (size_t)c->height * sizeof(int32_t));
    memcpy(e->vibc, c->vibc, (size_t)e->width * sizeof(int32_t));
    memcpy(e->rbv, c->rbv, (size_t)e->n_idx_cap * sizeof(int32_t));
    memcpy(e->cbv, c->cbv, (size_t)e->n_idx_cap * sizeof(int32_t));
    return JSLP_OK;
}

int jslp_engine_checkpoint_release(jslp_engine* e, int32_t id) {
    if (!e) return fail(JSLP_ERR_ARG, "checkpoint_release: null");
    if (id < 0 || id >= e->n_ck || !e->ck[id].live) return fail(JSLP_ERR_ARG, "checkpoint_release: no such checkpoint");
    checkpoint_free(&e->ck[id]);
    return JSLP_OK;
}

/* applyIncrementalCuts (incremental-branch-and-cut.ts:246-259), one node after the other */
int jslp_engine_relax_from(jslp_engine* e, int32_t checkpoint, int32_t n_nodes, const int32_t* cut_offsets,
                           const int8_t* type, const int32_t* var_index, const double* value, int check_cycles,
                           jslp_simplex_result* out, double* rhs, int32_t* var_index_by_row, int32_t out_stride) {
    if (checkpoint < 0)
        return jslp_engine_relax_batch(e, n_nodes, cut_offsets, type, var_index, value, check_cycles, out, rhs,
                                       var_index_by_row, out_stride);
    if (!e || n_nodes < 0 || !cut_offsets || !out) return fail(JSLP_ERR_ARG, "relax_from: null");
    if ((rhs || var_index_by_row) && out_stride < e->cap_rows) return fail(JSLP_ERR_ARG, "relax_from: out_stride < row capacity");
    for (int32_t i = 0; i < n_nodes; i++) {
        const int32_t a = cut_offsets[i], n = cut_offsets[i + 1] - a;
        int rc = jslp_engine_checkpoint_restore(e, checkpoint);
        if (rc) return rc;
        rc = jslp_engine_add_cuts(e, n, type ? type + a : 0, var_index ? var_index + a : 0, value ? value + a : 0);
        if (rc) return rc;
        rc = jslp_engine_simplex(e, check_cycles, &out[i]);
        if (rc) return rc;
        rc = jslp_engine_read_rhs(e, rhs ? rhs + (size_t)i * out_stride : 0,
                                  var_index_by_row ? var_index_by_row + (size_t)i * out_stride : 0);
        if (rc) return rc;
    }
    return JSLP_OK;
}

/* ---- ABI extras (host build buffer, compact read-back, work counters, device pool), mirrored sequentially ------------- */
int jslp_engine_host_matrix(jslp_engine* e, double** matrix, int64_t* n_doubles) {
    if (!e || !matrix) return fail(JSLP_ERR_ARG, "host_matrix: null");
    const size_t n = (size_t)e->height0 * e->width;
    if (!e->host_matrix) e->host_matrix = (double*)malloc(n * sizeof(double));
    if (!e->host_matrix) return fail(JSLP_ERR_NOMEM, "host_matrix: oom");
    memset(e->host_matrix, 0, n * sizeof(double));
    *matrix = e->host_matrix;
    if (n_doubles) *n_doubles = (int64_t)n;
    return JSLP_OK;
}

/* include/jslp_engine.h "outcomes left where they were computed": for this CPU stand-in device memory IS host memory and the raw
   state record is the result struct itself */
int32_t jslp_engine_state_record_bytes(void) { return (int32_t)sizeof(jslp_simplex_result); }
int jslp_engine_relax_batch_device(jslp_engine* e, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                                   const int32_t* var_index, const double* value, int check_cycles, void* d_states,
                                   double* d_rhs, int32_t* d_rows, int32_t row_stride) {
    if (!e || !d_states || !d_rhs || !d_rows) return fail(JSLP_ERR_ARG, "relax_batch_device: null pointer");
    return jslp_engine_relax_batch(e, n_nodes, cut_offsets, type, var_index, value, check_cycles, (jslp_simplex_result*)d_states,
                                   d_rhs, d_rows, row_stride);
}
int jslp_engine_results_from_states(jslp_engine* e, const void* states, int32_t n_nodes, jslp_simplex_result* out) {
    if (!e || !states || !out || n_nodes < 0) return fail(JSLP_ERR_ARG, "results_from_states: bad arguments");
    memcpy(out, states, sizeof(jslp_simplex_result) * (size_t)n_nodes);
    for (int32_t i = 0; i < n_nodes; i++) { out[i].cycle_start = 0; out[i].cycle_length = 0; }
    return JSLP_OK;
}

int jslp_engine_set_watched_variables(jslp_engine* e, const int32_t* var_indexes, int32_t n) {
    if (!e || !e->uploaded) return fail(JSLP_ERR_STATE, "set_watched_variables before upload");
    if (n < 0 || (n > 0 && !var_indexes)) return fail(JSLP_ERR_ARG, "set_watched_variables: bad arguments");
    for (int32_t i = 0; i < n; i++)
        if (var_indexes[i] < 0 || var_indexes[i] >= e->n_idx_cap) return fail(JSLP_ERR_ARG, "set_watched_variables: index out of range");
    free(e->watch);
    e->watch = 0;
    e->n_watch = 0;
    if (n > 0) {
        e->watch = (int32_t*)malloc((size_t)n * sizeof(int32_t));
        if (!e->watch) return fail(JSLP_ERR_NOMEM, "set_watched_variables: oom");
        memcpy(e->watch, var_indexes, (size_t)n * sizeof(int32_t));
        e->n_watch = n;
    }
    return JSLP_OK;
}

int jslp_engine_relax_watched(jslp_engine* e, int32_t n_cuts, const int8_t* type, const int32_t* var_index,
                              const double* value, int check_cycles, jslp_simplex_result* out, int32_t* watched_row,
                              double* watched_value) {
    if (!e) return fail(JSLP_ERR_ARG, "relax_watched: null");
    if (e->n_watch <= 0 || e->n_watch > e->cap_rows)
        return fail(JSLP_ERR_ARG, "relax_watched: one node at a time, after set_watched_variables (at most row_capacity of them)");
    int rc = jslp_engine_relax(e, n_cuts, type, var_index, value, check_cycles, out, 0, 0);
    if (rc) return rc;
    for (int32_t i = 0; i < e->n_watch; i++) { /* rowByVarIndex[v], then matrix[row * width + rhsColumn] (mip-utils.ts:43-61) */
        const int32_t r = e->rbv[e->watch[i]];
        const int basic = r > 0 && r < e->height;
        if (watched_row) watched_row[i] = basic ? r : -1;
        if (watched_value) watched_value[i] = basic ? e->matrix[(size_t)r * e->width] : 0.0;
    }
    return JSLP_OK;
}

int jslp_engine_relax_batch_watched(jslp_engine* e, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                                    const int32_t* var_index, const double* value, int check_cycles,
                                    jslp_simplex_result* out, int32_t* watched_row, double* watched_value) {
    if (!e || n_nodes < 0 || !cut_offsets || !out) return fail(JSLP_ERR_ARG, "relax_batch_watched: null");
    if (e->n_watch <= 0 || e->n_watch > e->cap_rows)
        return fail(JSLP_ERR_ARG, "relax_batch_watched: after set_watched_variables (at most row_capacity of them)");
    for (int32_t i = 0; i < n_nodes; i++) {
        const int32_t a = cut_offsets[i], n = cut_offsets[i + 1] - a;
        int rc = jslp_engine_relax_watched(e, n, type ? type + a : 0, var_index ? var_index + a : 0, value ? value + a : 0, check_cycles,
                                           &out[i], watched_row ? watched_row + (size_t)i * e->n_watch : 0,
                                           watched_value ? watched_value + (size_t)i * e->n_watch : 0);
        if (rc) return rc;
    }
    return JSLP_OK;
}

/* zero-copy variant of the ABI: here backed by heap buffers */
int jslp_engine_relax_batch_watched_pinned(jslp_engine* e, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                                           const int32_t* var_index, const double* value, int check_cycles,
                                           jslp_simplex_result* out, const int32_t** watched_row, const double** watched_value) {
    if (!e) return fail(JSLP_ERR_ARG, "relax_batch_watched_pinned: null");
    static __thread double* b_val = 0;
    static __thread int32_t* b_row = 0;
    static __thread size_t b_cap = 0;
    const size_t need = (size_t)(n_nodes > 0 ? n_nodes : 1) * (size_t)(e->n_watch > 0 ? e->n_watch : 1);
    if (need > b_cap) {
        b_val = (double*)realloc(b_val, need * sizeof(double));
        b_row = (int32_t*)realloc(b_row, need * sizeof(int32_t));
        b_cap = need;
    }
    int rc = jslp_engine_relax_batch_watched(e, n_nodes, cut_offsets, type, var_index, value, check_cycles, out, b_row, b_val);
    if (rc) return rc;
    if (watched_row) *watched_row = b_row;
    if (watched_value) *watched_value = b_val;
    return JSLP_OK;
}

/* the compact read-back in "device" memory (host memory here; the raw state record is the result struct, see relax_batch_device) */
int jslp_engine_relax_batch_watched_device(jslp_engine* e, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                                           const int32_t* var_index, const double* value, int check_cycles, void* d_states,
                                           int32_t* d_watched_row, double* d_watched_value) {
    if (!e || !d_states || !d_watched_row || !d_watched_value) return fail(JSLP_ERR_ARG, "relax_batch_watched_device: null pointer");
    return jslp_engine_relax_batch_watched(e, n_nodes, cut_offsets, type, var_index, value, check_cycles, (jslp_simplex_result*)d_states,
                                           d_watched_row, d_watched_value);
}
int32_t jslp_engine_watched_count(const jslp_engine* e) { return e ? e->n_watch : 0; }

int jslp_engine_set_counting(jslp_engine* e, int enabled) {
    if (!e) return fail(JSLP_ERR_ARG, "set_counting: null");
    e->counting = enabled ? 1 : 0;
    memset(&e->wc, 0, sizeof e->wc);
    return JSLP_OK;
}

int jslp_engine_get_counters(jslp_engine* e, jslp_work_counters* out) {
    if (!e || !out) return fail(JSLP_ERR_ARG, "get_counters: null");
    *out = e->wc;
    return JSLP_OK;
}

struct jslp_pool {
    int32_t n;
    jslp_engine** members; /* members[0] = the primary (not owned) */
    unsigned long long synced_seq;
    int synced;
    double* b_rhs;
    int32_t* b_rows;
    size_t b_cap;
};

int jslp_pool_size(const jslp_pool* p) { return p ? p->n : 0; }

void jslp_pool_destroy(jslp_pool* p) {
    if (!p) return;
    for (int32_t i = 1; i < p->n; i++) jslp_engine_destroy(p->members[i]);
    free(p->members); free(p->b_rhs); free(p->b_rows);
    free(p);
}

int jslp_pool_create(jslp_pool** out, jslp_engine* primary, const int32_t* devices, int32_t n_devices) {
    if (!out || !primary || !devices || n_devices < 1) return fail(JSLP_ERR_ARG, "pool_create: bad arguments");
    jslp_pool* p = (jslp_pool*)calloc(1, sizeof *p);
    if (!p) return fail(JSLP_ERR_NOMEM, "pool_create: oom");
    p->members = (jslp_engine**)calloc((size_t)n_devices, sizeof(jslp_engine*));
    if (!p->members) { free(p); return fail(JSLP_ERR_NOMEM, "pool_create: oom"); }
    p->members[0] = primary;
    p->n = 1;
    for (int32_t i = 1; i < n_devices; i++) {
        int rc = jslp_engine_create(&p->members[i], devices[i], primary->height0, primary->width, primary->cap_rows, primary->precision);
        if (rc) { jslp_pool_destroy(p); return rc; }
        p->n = i + 1;
    }
    *out = p;
    return JSLP_OK;
}

/* every member receives the primary's saved root (savedState, backup.ts:13-51) and restores it */
int jslp_pool_sync_root(jslp_pool* p) {
    if (!p) return fail(JSLP_ERR_ARG, "pool_sync_root: null");
    jslp_engine* e = p->members[0];
    if (!e->uploaded || !e->has_save) return fail(JSLP_ERR_STATE, "pool_sync_root: the primary has no saved root (save() first)");
    for (int32_t i = 1; i < p->n; i++) {
        jslp_engine* m = p->members[i];
        const size_t cells = (size_t)e->s_height * e->width;
        memcpy(m->s_matrix, e->s_matrix, cells * sizeof(double));
        memcpy(m->s_vibr, e->s_vibr, (size_t)e->s_height * sizeof(int32_t));
        memcpy(m->s_vibc, e->s_vibc, (size_t)e->width * sizeof(int32_t));
        memcpy(m->s_rbv, e->s_rbv, (size_t)e->n_idx_cap * sizeof(int32_t));
        memcpy(m->s_cbv, e->s_cbv, (size_t)e->n_idx_cap * sizeof(int32_t));
        memcpy(m->unrestricted, e->unrestricted, (size_t)e->n_idx_cap);
        if (e->is_int) {
            if (!m->is_int) m->is_int = (uint8_t*)calloc((size_t)e->n_idx_cap, 1);
            if (!m->is_int) return fail(JSLP_ERR_NOMEM, "pool_sync_root: oom");
            memcpy(m->is_int, e->is_int, (size_t)e->n_idx_cap);
        }
        if (e->n_opt > 0) {
            const size_t nb = (size_t)e->n_opt * e->width * sizeof(double);
            free(m->oo); free(m->s_oo); free(m->defer);
            m->oo = (double*)malloc(nb);
            m->s_oo = (double*)malloc(nb);
            m->defer = (int32_t*)calloc((size_t)e->width * 2, sizeof(int32_t));
            if (!m->oo || !m->s_oo || !m->defer) return fail(JSLP_ERR_NOMEM, "pool_sync_root: oom");
            memcpy(m->s_oo, e->s_oo, nb);
        }
        m->n_opt = e->n_opt;
        m->s_height = e->s_height;
        m->s_last_element_index = e->s_last_element_index;
        m->has_save = 1;
        m->uploaded = 1;
        m->evaluation = e->evaluation;
        m->feasible = 1;
        m->bounded = 1;
        checkpoints_clear(m);
        int rc = jslp_engine_restore(m);
        if (rc) return rc;
    }
    p->synced_seq = e->root_seq;
    p->synced = 1;
    return JSLP_OK;
}

int jslp_pool_relax_batch(jslp_pool* p, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                          const int32_t* var_index, const double* value, int check_cycles, jslp_simplex_result* out,
                          double* rhs, int32_t* var_index_by_row, int32_t out_stride) {
    if (!p || n_nodes < 0 || !cut_offsets || !out) return fail(JSLP_ERR_ARG, "pool_relax_batch: null");
    jslp_engine* e = p->members[0];
    if (!e->uploaded || !e->has_save) return fail(JSLP_ERR_STATE, "pool_relax_batch: the primary has no saved root (save() first)");
    if ((rhs || var_index_by_row) && out_stride < e->cap_rows) return fail(JSLP_ERR_ARG, "pool_relax_batch: out_stride < row capacity");
    if (!p->synced || p->synced_seq != e->root_seq) {
        int rc = jslp_pool_sync_root(p);
        if (rc) return rc;
    }
    for (int32_t mi = 0; mi < p->n; mi++) { /* the same contiguous ranges the product hands its members */
        const int32_t first = (int32_t)((int64_t)n_nodes * mi / p->n), last = (int32_t)((int64_t)n_nodes * (mi + 1) / p->n);
        for (int32_t i = first; i < last; i++) {
            const int32_t a = cut_offsets[i], n = cut_offsets[i + 1] - a;
            int rc = jslp_engine_relax(p->members[mi], n, type ? type + a : 0, var_index ? var_index + a : 0, value ? value + a : 0,
                                       check_cycles, &out[i], rhs ? rhs + (size_t)i * out_stride : 0,
                                       var_index_by_row ? var_index_by_row + (size_t)i * out_stride : 0);
            if (rc) return rc;
        }
    }
    return JSLP_OK;
}

int jslp_pool_relax_batch_pinned(jslp_pool* p, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                                 const int32_t* var_index, const double* value, int check_cycles,
                                 jslp_simplex_result* out, const double** rhs, const int32_t** var_index_by_row,
                                 int32_t* out_stride) {
    if (!p) return fail(JSLP_ERR_ARG, "pool_relax_batch_pinned: null");
    const int32_t cap = p->members[0]->cap_rows;
    const size_t need = (size_t)(n_nodes > 0 ? n_nodes : 1) * cap;
    if (need > p->b_cap) {
        p->b_rhs = (double*)realloc(p->b_rhs, need * sizeof(double));
        p->b_rows = (int32_t*)realloc(p->b_rows, need * sizeof(int32_t));
        p->b_cap = need;
    }
    int rc = jslp_pool_relax_batch(p, n_nodes, cut_offsets, type, var_index, value, check_cycles, out, p->b_rhs, p->b_rows, cap);
    if (rc) return rc;
    if (rhs) *rhs = p->b_rhs;
    if (var_index_by_row) *var_index_by_row = p->b_rows;
    if (out_stride) *out_stride = cap;
    return JSLP_OK;
}

/* the compact read-back over the pool: rowByVarIndex / RHS cell of the watched variables per node (mip-utils.ts:43-61) */
int32_t jslp_pool_watched_count(const jslp_pool* p) {
    if (!p || p->n <= 0) return 0;
    for (int32_t i = 1; i < p->n; i++)
        if (p->members[i]->n_watch != p->members[0]->n_watch) return -1;
    return p->members[0]->n_watch;
}

int jslp_pool_set_watched_variables(jslp_pool* p, const int32_t* var_indexes, int32_t n) {
    if (!p) return fail(JSLP_ERR_ARG, "pool_set_watched_variables: null");
    jslp_engine* e = p->members[0];
    if (e->uploaded && e->has_save && (!p->synced || p->synced_seq != e->root_seq)) {
        int rc = jslp_pool_sync_root(p);
        if (rc) return rc;
    }
    for (int32_t i = 0; i < p->n; i++) {
        int rc = jslp_engine_set_watched_variables(p->members[i], var_indexes, n);
        if (rc) return rc;
    }
    return JSLP_OK;
}

int jslp_pool_relax_batch_watched(jslp_pool* p, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                                  const int32_t* var_index, const double* value, int check_cycles, jslp_simplex_result* out,
                                  int32_t* watched_row, double* watched_value) {
    if (!p || n_nodes < 0 || !cut_offsets || !out) return fail(JSLP_ERR_ARG, "pool_relax_batch_watched: null");
    jslp_engine* e = p->members[0];
    if (!e->uploaded || !e->has_save) return fail(JSLP_ERR_STATE, "pool_relax_batch: the primary has no saved root (save() first)");
    if (e->n_watch <= 0) return fail(JSLP_ERR_ARG, "pool_relax_batch_watched: after jslp_pool_set_watched_variables");
    for (int32_t i = 0; i < p->n; i++)
        if (p->members[i]->n_watch != e->n_watch) return fail(JSLP_ERR_STATE, "pool_relax_batch_watched: the members' watched variables differ from the primary's (jslp_pool_set_watched_variables sets them all)");
    if (!p->synced || p->synced_seq != e->root_seq) {
        int rc = jslp_pool_sync_root(p);
        if (rc) return rc;
    }
    const size_t nw = (size_t)e->n_watch;
    for (int32_t mi = 0; mi < p->n; mi++) {
        const int32_t first = (int32_t)((int64_t)n_nodes * mi / p->n), last = (int32_t)((int64_t)n_nodes * (mi + 1) / p->n);
        for (int32_t i = first; i < last; i++) {
            const int32_t a = cut_offsets[i], n = cut_offsets[i + 1] - a;
            int rc = jslp_engine_relax_watched(p->members[mi], n, type ? type + a : 0, var_index ? var_index + a : 0, value ? value + a : 0,
                                               check_cycles, &out[i], watched_row ? watched_row + (size_t)i * nw : 0,
                                               watched_value ? watched_value + (size_t)i * nw : 0);
            if (rc) return rc;
        }
    }
    return JSLP_OK;
}

int jslp_pool_relax_batch_watched_pinned(jslp_pool* p, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                                         const int32_t* var_index, const double* value, int check_cycles, jslp_simplex_result* out,
                                         const int32_t** watched_row, const double** watched_value) {
    if (!p) return fail(JSLP_ERR_ARG, "pool_relax_batch_watched_pinned: null");
    const size_t nw = (size_t)(p->members[0]->n_watch > 0 ? p->members[0]->n_watch : 1);
    const size_t need = (size_t)(n_nodes > 0 ? n_nodes : 1) * nw;
    if (need > p->b_cap) {
        p->b_rhs = (double*)realloc(p->b_rhs, need * sizeof(double));
        p->b_rows = (int32_t*)realloc(p->b_rows, need * sizeof(int32_t));
        p->b_cap = need;
    }
    int rc = jslp_pool_relax_batch_watched(p, n_nodes, cut_offsets, type, var_index, value, check_cycles, out, p->b_rows, p->b_rhs);
    if (rc) return rc;
    if (watched_row) *watched_row = p->b_rows;
    if (watched_value) *watched_value = p->b_rhs;
    return JSLP_OK;
}

int jslp_pool_set_counting(jslp_pool* p, int enabled) {
    if (!p) return fail(JSLP_ERR_ARG, "pool_set_counting: null");
    for (int32_t i = 0; i < p->n; i++) jslp_engine_set_counting(p->members[i], enabled);
    return JSLP_OK;
}

int jslp_pool_get_counters(jslp_pool* p, jslp_work_counters* out) {
    if (!p || !out) return fail(JSLP_ERR_ARG, "pool_get_counters: null");
    memset(out, 0, sizeof *out);
    for (int32_t i = 0; i < p->n; i++) {
        const jslp_work_counters* c = &p->members[i]->wc;
        out->relaxations += c->relaxations; out->simplex_calls += c->simplex_calls; out->pivots += c->pivots;
        out->gated_cells += c->gated_cells; out->gated_rows += c->gated_rows; out->restored_rows += c->restored_rows;
        out->cut_rows += c->cut_rows; out->height_sum += c->height_sum;
        /* (resident_*: the sequential restatement has no register-resident kernels; always 0) */
    }
    return JSLP_OK;
}

int jslp_engine_dims(const jslp_engine* e, int32_t* height, int32_t* width, int32_t* n_var_indexes) {
    if (!e) return fail(JSLP_ERR_ARG, "dims: null");
    if (height) *height = e->height;
    if (width) *width = e->width;
    if (n_var_indexes) *n_var_indexes = e->n_idx_cap;
    return JSLP_OK;
}

int jslp_engine_download(jslp_engine* e, double* matrix, int32_t* var_index_by_row, int32_t* var_index_by_col,
                         int32_t* row_by_var_index, int32_t* col_by_var_index) {
    if (!e || !e->uploaded) return fail(JSLP_ERR_STATE, "download before upload");
    if (matrix) memcpy(matrix, e->matrix, (size_t)e->height * e->width * sizeof(double));
    if (var_index_by_row) memcpy(var_index_by_row, e->vibr, (size_t)e->height * sizeof(int32_t));
    if (var_index_by_col) memcpy(var_index_by_col, e->vibc, (size_t)e->width * sizeof(int32_t));
    if (row_by_var_index) memcpy(row_by_var_index, e->rbv, (size_t)e->n_idx_cap * sizeof(int32_t));
    if (col_by_var_index) memcpy(col_by_var_index, e->cbv, (size_t)e->n_idx_cap * sizeof(int32_t));
    return JSLP_OK;
}

int jslp_engine_pivot_trace(jslp_engine* e, int32_t* row_col, int64_t max_pairs, int64_t* n_pivots) {
    if (!e || !n_pivots) return fail(JSLP_ERR_ARG, "pivot_trace: null");
    *n_pivots = e->n_trace;
    if (row_col && max_pairs > 0) {
        int64_t n = e->n_trace < max_pairs ? e->n_trace : max_pairs;
        memcpy(row_col, e->trace, (size_t)n * 2 * sizeof(int32_t));
    }
    return JSLP_OK;
}

const char* jslp_engine_last_path(const jslp_engine* e) { (void)e; return "oracle"; }

int jslp_engine_set_timing(jslp_engine* e, int enabled) { (void)e; (void)enabled; return JSLP_OK; }
int jslp_engine_get_timing(jslp_engine* e, double* update_kernel_ms, int64_t* update_kernel_launches,
                           double* total_device_ms) {
    (void)e;
    if (update_kernel_ms) *update_kernel_ms = 0;
    if (update_kernel_launches) *update_kernel_launches = 0;
    if (total_device_ms) *total_device_ms = 0;
    return JSLP_OK;
}
