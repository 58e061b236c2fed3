/*
 * jslp_engine.h -- C ABI of the MI355X dense-tableau simplex engine.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch / no C++ types.  One
 * `jslp_engine` is the device-resident state of ONE reference `Tableau` (src/tableau/tableau.ts:46-99):
 * the fp64 row-major matrix (row 0 = reduced costs, column 0 = RHS), the four index maps, the root
 * snapshot used by branch-and-bound and the scratch the kernels need.  Every entry point replaces the
 * reference method cited next to it and keeps its semantics bit for bit (no FMA contraction, IEEE
 * division, first-index tie-breaks).
 *
 * Two libraries export exactly these symbols:
 *   jslpsolver_amd/csrc/libjslp_hip.so   -- the product: hand-written HIP kernels for gfx950
 *   oracle/libjslp_oracle.so             -- TEST ONLY: sequential C restatement of the reference
 * All functions are synchronous for the caller (the reference is single-threaded and synchronous,
 * src/main.ts:94-147); asynchrony (streams, chunked launches, per-node workgroups) stays inside.
 *
 * Error convention: the reference's hot path never throws -- outcomes are flags (simplex.ts:73-76,
 * 86-92, 298-303).  So every function returns an int status (0 = OK, <0 = the call itself failed) and
 * solver outcomes travel in jslp_simplex_result.
 */
#ifndef JSLP_ENGINE_H
#define JSLP_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JSLP_OK 0
#define JSLP_ERR_ARG (-1)         /* bad argument (null pointer, index out of range, size mismatch)  */
#define JSLP_ERR_DEVICE (-2)      /* HIP runtime error or no gfx950 device; see jslp_last_error()      */
#define JSLP_ERR_NOMEM (-3)
#define JSLP_ERR_STATE (-4)       /* call order violated (e.g. simplex before upload)                  */
#define JSLP_ERR_CAPACITY (-5)    /* add_cuts beyond row_capacity, pivot history beyond its limit      */
#define JSLP_ERR_UNSUPPORTED (-6) /* feature outside the built scope                                   */

/* cut direction, reference BranchCut.type (src/tableau/types.ts:17-21) */
#define JSLP_CUT_MIN 0 /* "min": x >= value  (sign -1 in cutting-strategies.ts:41) */
#define JSLP_CUT_MAX 1 /* "max": x <= value  (sign +1) */

typedef struct jslp_engine jslp_engine;

/* Outcome of one Tableau.simplex() call (src/tableau/simplex.ts:14-23). */
typedef struct jslp_simplex_result {
    int32_t feasible;            /* tableau.feasible: 0 when phase 1 fails OR a cycle is detected          */
    int32_t bounded;             /* tableau.bounded (simplex.ts:15,300)                                     */
    int32_t optimal;             /* 1 iff phase 2 hit "no entering column": setEvaluation() ran and
                                    simplexIters += 1 (simplex.ts:265-269)                                  */
    int32_t unbounded_var_index; /* tableau.unboundedVarIndex (simplex.ts:301) or -1                        */
    int32_t pivots_phase1;       /* return value of phase1() (simplex.ts:53,75,91)                          */
    int32_t pivots_phase2;       /* return value of phase2(); -1 when phase 2 did not run                   */
    int32_t cycle_phase;         /* 0 = none, 1 / 2 = "Cycle in phase N" (simplex.ts:86,313)                */
    int32_t cycle_start;         /* checkForCycles() result [start, length] (simplex.ts:415-440)            */
    int32_t cycle_length;
    int32_t height;              /* current number of rows (grows with cuts)                                */
    double obj_cell;             /* raw matrix[0] after the call                                            */
    double evaluation;           /* when optimal: round((EPS + matrix[0]) * rc) / rc, rc = round(1/precision)
                                    (tableau.ts:420-426); -Infinity when unbounded; otherwise the previous
                                    value is kept, exactly like tableau.evaluation                          */
} jslp_simplex_result;

/* Library identity: "hip-gfx950" for the product, "oracle-c" for the test restatement. */
const char* jslp_backend_name(void);
/* Human-readable text of the last failure on this thread ("" if none). */
const char* jslp_last_error(void);
/* Number of usable devices (0 for the oracle / when no GPU is visible). */
int jslp_device_count(void);

/*
 * Threading: an engine belongs to one host thread at a time (the reference is single-threaded); different engines may
 * be driven from different threads.  destroy() parks the engine's stream, events, pinned staging buffers and device
 * arenas in a small per-process pool that the next create() on the same device takes over (a Solve of a small model
 * is otherwise dominated by ~4 ms of resource set-up and tear-down); jslp_release_pooled_resources() frees whatever is
 * parked, e.g. before unloading the library.
 */
void jslp_release_pooled_resources(void);

/*
 * new Tableau(precision) + Tableau.initialize(width, height, ...) (tableau.ts:94-99, 292-317).
 * row_capacity >= height bounds how many cut rows add_cuts may append (the reference reallocates in
 * cutting-strategies.ts:24-30; device memory is sized once instead).  device = HIP device ordinal.
 */
int jslp_engine_create(jslp_engine** out, int device, int32_t height, int32_t width, int32_t row_capacity,
                       double precision);
void jslp_engine_destroy(jslp_engine* e);

/*
 * Hand over the tableau built by Tableau._resetMatrix (tableau.ts:319-380): matrix is height*width
 * doubles, row-major, stride width (the reference layout, tableau.ts:49-54,304).  var_index_by_row[0] and
 * var_index_by_col[0] are -1.  unrestricted_var_indexes lists model.unrestrictedVariables keys
 * (model.ts:181-183).  The element-index counter continues at width+height-2 (tableau.ts:312-316).
 * Resets snapshot, evaluation and pivot counters.
 */
int jslp_engine_upload(jslp_engine* e, const double* matrix, const int32_t* var_index_by_row,
                       const int32_t* var_index_by_col, const int32_t* unrestricted_var_indexes,
                       int32_t n_unrestricted);

/*
 * Optional objectives (soft constraints): the priority-ordered extra cost rows `optionalObjectives[o].reducedCosts`
 * (tableau.ts:71, 278-290; built by _resetMatrix :335-338).  rows = n x width doubles, objective o at rows + o*width,
 * already sorted by ascending priority; column 0 is the objective's evaluation cell (branch-and-cut.ts:107-127 reads it).
 * pivot() updates them (simplex.ts:394-412), phase 2 consults them when no column prices out on the main cost row
 * (simplex.ts:155-162, 221-263), save()/restore() snapshot them (backup.ts:37-43, 94-104).  Call after upload().
 */
int jslp_engine_set_optional_objectives(jslp_engine* e, int32_t n, const double* rows);
/* Current optional-objective rows (n x width, as above); n_out receives n. */
int jslp_engine_get_optional_objectives(jslp_engine* e, double* rows, int32_t* n_out);

/* Tableau.simplex(): phase1 then phase2 (simplex.ts:14-23).  check_cycles = model.checkForCycles. */
int jslp_engine_simplex(jslp_engine* e, int check_cycles, jslp_simplex_result* out);

/* Tableau.pivot(r, c) on its own (simplex.ts:330-413); used by putInBase/takeOutOfBase and by tests. */
int jslp_engine_pivot(jslp_engine* e, int32_t row, int32_t col);

/* Tableau.save(): device-resident snapshot of matrix + index maps + counters (backup.ts:13-51). */
int jslp_engine_save(jslp_engine* e);
/* Tableau.restore(): no-op before the first save (backup.ts:53-105). */
int jslp_engine_restore(jslp_engine* e);

/*
 * Tableau.addCutConstraints(cuts) (cutting-strategies.ts:16-72): appends n rows; each new row gets the
 * next element index exactly as getNewElementIndex() (tableau.ts:393-401).
 */
int jslp_engine_add_cuts(jslp_engine* e, int32_t n, const int8_t* type, const int32_t* var_index,
                         const double* value);

/*
 * One LP relaxation = BranchAndCutService.applyCuts (branch-and-cut.ts:33-37): restore + add_cuts +
 * simplex, then the read-back branch-and-bound needs (mip-utils.ts:43-61,100-126): the RHS column and the
 * row -> variable map, `height` entries each (rhs / var_index_by_row may be NULL to skip).
 */
int jslp_engine_relax(jslp_engine* e, int32_t n_cuts, const int8_t* type, const int32_t* var_index,
                      const double* value, int check_cycles, jslp_simplex_result* out, double* rhs,
                      int32_t* var_index_by_row);

/*
 * The same for a batch of independent branch-and-bound nodes (each a pure function of the saved root and
 * its cut list): node i owns cuts [cut_offsets[i], cut_offsets[i+1]).  Results land in
 * out[i]; rhs / var_index_by_row are n_nodes x out_stride arrays (row i = node i, first out[i].height
 * entries valid).  Afterwards the engine's own (live) tableau holds ONE of the batch's nodes -- which one is
 * unspecified (the nodes are spread over tableau copies, the live one being the first of them): restore() before
 * using it; the engine's evaluation is the last node's.  This is the unit that shards across GPUs (one engine per
 * rank, disjoint node ranges).
 */
int jslp_engine_relax_batch(jslp_engine* e, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                            const int32_t* var_index, const double* value, int check_cycles,
                            jslp_simplex_result* out, double* rhs, int32_t* var_index_by_row,
                            int32_t out_stride);

/*
 * Zero-copy variant: the outcomes stay in the engine's own pinned read-back buffer.  On return *rhs / *var_index_by_row
 * point at n_nodes x *out_stride arrays (row i = node i, first out[i].height entries valid) that remain valid until
 * the next call on this engine; pass NULL for what is not needed (less PCIe traffic).  Same semantics otherwise.
 * *out_stride is the engine's choice (>= the row capacity; padded so that every node's slice is 16-byte aligned).
 */
int jslp_engine_relax_batch_pinned(jslp_engine* e, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                                   const int32_t* var_index, const double* value, int check_cycles,
                                   jslp_simplex_result* out, const double** rhs, const int32_t** var_index_by_row,
                                   int32_t* out_stride);

/*
 * Outcomes left where they were computed (the one-process-per-GPU path, jslpsolver_amd/sharding.py: the exchange step of a batch
 * sharded over ranks is an RCCL all-gather, whose input must be device memory -- SURVEY.md 8e; replaces the pinned-host ->
 * device bounce of that path).  relax_batch_device = jslp_engine_relax_batch whose per-node outcome -- a raw state record of
 * jslp_engine_state_record_bytes() bytes, the RHS column and the row map (row_stride entries per node, row_stride >= the row
 * capacity; a multiple of 4 keeps the kernels on their 16-byte stores) -- is written into memory of the ENGINE'S DEVICE that the
 * caller owns; nothing crosses PCIe.  results_from_states turns records (in host memory; this rank's or, after the exchange,
 * any rank's) into the result structs of the other entry points; a record carries no pivot history, so cycle_start /
 * cycle_length come back 0 (flags and cycle_phase are exact).  The oracle library implements the same three entry points over
 * plain host memory.
 */
int32_t jslp_engine_state_record_bytes(void);
int jslp_engine_relax_batch_device(jslp_engine* e, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                                   const int32_t* var_index, const double* value, int check_cycles, void* d_states,
                                   double* d_rhs, int32_t* d_rows, int32_t row_stride);
int jslp_engine_results_from_states(jslp_engine* e, const void* states, int32_t n_nodes, jslp_simplex_result* out);

/*
 * MIR cuts (options.useMIRCuts, src/model.ts:354-356).  set_integer_variables hands over model.integerVariables' indexes:
 * the `variable.isInteger` test of addLowerBoundMIRCut (src/tableau/cutting-strategies.ts:82-85, 120-121); call after
 * upload().  apply_mir_cuts = Tableau.applyMIRCuts() (cutting-strategies.ts:199-212): scans rows 1..height-1 in order
 * and appends a lower-bound MIR cut (:74-135) for the first <= 10 rows whose basic variable is an integer variable with
 * a fractional value; each new row gets the next element index (tableau.ts:393-401).  *n_added receives the number of
 * rows appended.  mir_round = one turn of the services' MIR loop (branch-and-cut.ts:41-43): applyMIRCuts() + simplex()
 * + the read-back of jslp_engine_relax (the host computes computeFractionalVolume, mip-utils.ts:67-92, from it).
 */
int jslp_engine_set_integer_variables(jslp_engine* e, const int32_t* var_indexes, int32_t n);
int jslp_engine_apply_mir_cuts(jslp_engine* e, int32_t* n_added);
int jslp_engine_mir_round(jslp_engine* e, int check_cycles, int32_t* n_added, jslp_simplex_result* out, double* rhs,
                          int32_t* var_index_by_row);

/*
 * fp32-vs-fp64 tolerance sweep.  Runs simplex() on an fp32 COPY of the live tableau with the
 * tolerance `precision` (the reference's `precision`, tableau.ts:96) and returns what jslp_engine_relax returns (the RHS
 * column widened back to double); *device_ms receives the device time of the pivot loop.  Same pivoting rules, same
 * kernels compiled for float; the fp64 state of the engine is NOT modified and no pivot is traced.  This is an
 * experiment, not a drop-in path: fp32 has no reference to be exact against, the caller compares flags / objective /
 * pivot counts with the fp64 solve.  The test library answers JSLP_ERR_UNSUPPORTED.
 */
int jslp_engine_simplex_f32(jslp_engine* e, double precision, int check_cycles, jslp_simplex_result* out, double* rhs,
                            int32_t* var_index_by_row, double* device_ms);

/*
 * Device-resident checkpoints: the StateCheckpoint of the incremental branch-and-bound service
 * (src/tableau/incremental-branch-and-cut.ts:31-44).  createCheckpoint (:55-70) copies the live matrix, the four index
 * maps, height and lastElementIndex; restoreCheckpoint (:72-107) puts them back.  Exactly like the reference, a
 * checkpoint does NOT carry the optional-objective rows or the saved root (restoring one leaves both untouched); the
 * host keeps `evaluation` / `feasible` (:42-43) next to the id, the engine remembers `evaluation` for the "kept when not
 * optimal" rule.  Checkpoints live in HBM (H x W x 8 bytes each; 288 GB hold thousands) until released, the next
 * upload() or destroy().
 */
int jslp_engine_checkpoint_create(jslp_engine* e, int32_t* id_out);
int jslp_engine_checkpoint_restore(jslp_engine* e, int32_t id);
int jslp_engine_checkpoint_release(jslp_engine* e, int32_t id);

/*
 * applyIncrementalCuts' fast path (incremental-branch-and-cut.ts:248-253) for the children of one parent: every node
 * = restoreCheckpoint(checkpoint) + addCutConstraints(its cuts, normally the single new cut) + simplex() + the
 * read-back of jslp_engine_relax_batch.  checkpoint = -1 starts from the saved root instead (= relax_batch, the
 * fallback path :254-258).  The live tableau is left holding the LAST node.
 */
int jslp_engine_relax_from(jslp_engine* e, int32_t checkpoint, int32_t n_nodes, const int32_t* cut_offsets,
                           const int8_t* type, const int32_t* var_index, const double* value, int check_cycles,
                           jslp_simplex_result* out, double* rhs, int32_t* var_index_by_row, int32_t out_stride);

/*
 * Host-side build straight into pinned memory (model.ts:278-419 + tableau.ts:319-380 write the dense
 * tableau cell by cell).  host_matrix hands the host a zero-filled height x width fp64 buffer in pinned (DMA-able) memory
 * owned by the engine -- the binding makes it the `Float64Array` behind `tableau.matrix` before _resetMatrix runs
 * (tableau.ts:304 allocates that array; a fresh Float64Array is zero-filled, so is this buffer on every call).  Passing
 * that very pointer to jslp_engine_upload() turns the upload into one DMA from pinned memory (no pageable staging copy).
 * The buffer stays valid until destroy(); *n_doubles receives height * width.
 */
int jslp_engine_host_matrix(jslp_engine* e, double** matrix, int64_t* n_doubles);

/*
 * Compact read-back for the branch-and-bound tree.  Between relaxations the host reads only the integer variables
 * (isIntegral / getMostFractionalVar, mip-utils.ts:43-61, 100-126: `rowByVarIndex[v]` then `matrix[row * width + rhsColumn]`).
 * set_watched_variables lists those variable indexes once; relax_watched = jslp_engine_relax whose read-back is, per
 * watched variable i, watched_row[i] (rowByVarIndex, -1 when not basic) and watched_value[i] (its RHS cell; 0 when not
 * basic) -- n_watched x 12 bytes instead of height x 12.  jslp_engine_read_rhs completes the picture when the tree is done.
 */
int jslp_engine_set_watched_variables(jslp_engine* e, const int32_t* var_indexes, int32_t n);
int jslp_engine_relax_watched(jslp_engine* e, int32_t n_cuts, const int8_t* type, const int32_t* var_index,
                              const double* value, int check_cycles, jslp_simplex_result* out, int32_t* watched_row,
                              double* watched_value);
/*
 * relax_batch with the compact read-back: node i's watched variables at watched_row / watched_value [i * n_watched ..
 * (i + 1) * n_watched).  A large batch of full read-backs is bound by the PCIe link (height x 12 bytes per node: 27.8 MB for
 * the 2416 Monster_II nodes), not by the GPU; this is what a host that walks the tree itself needs per node (the full
 * RHS column of an incumbent is one jslp_engine_relax away).  Both outputs may be NULL.  Same preconditions as relax_batch.
 */
int jslp_engine_relax_batch_watched(jslp_engine* e, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                                    const int32_t* var_index, const double* value, int check_cycles,
                                    jslp_simplex_result* out, int32_t* watched_row, double* watched_value);
/* the same without the copy: *watched_row / *watched_value point into the engine's pinned read-back buffer (n_nodes x
 * n_watched each), valid until the next call on this engine */
int jslp_engine_relax_batch_watched_pinned(jslp_engine* e, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                                           const int32_t* var_index, const double* value, int check_cycles,
                                           jslp_simplex_result* out, const int32_t** watched_row, const double** watched_value);
/* the compact read-back left in DEVICE memory of the caller (round 5): per node the raw state record (jslp_engine_state_record_bytes()
 * bytes, see jslp_engine_relax_batch_device), watched_row and watched_value -- n_watched x 12 + 128 bytes per node instead of
 * row_stride x 12 + 128.  This IS the payload of the multi-process exchange (jslpsolver_amd/sharding.py all-gathers it over RCCL: ~1.5 KB
 * per Monster_II node instead of 11.4 KB): what every rank's copy of the tree reads between relaxations (mip-utils.ts:43-61,100-126);
 * the full column of the one leaf the tree commits to is re-evaluated on demand (jslp_engine_relax), as the JS host's flush() does. */
int jslp_engine_relax_batch_watched_device(jslp_engine* e, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                                           const int32_t* var_index, const double* value, int check_cycles, void* d_states,
                                           int32_t* d_watched_row, double* d_watched_value);
/* how many variables jslp_engine_set_watched_variables registered last (0: none) -- what a binding sizes the compact outputs with */
int32_t jslp_engine_watched_count(const jslp_engine* e);

/*
 * Work counters (bench.py's roofline of the relaxation path): what the calls since the last reset really had
 * to touch, counted by the kernels themselves when counting is on (off by default: it costs a reduction per pivot).
 * gated_cells = sum over pivots of (rows passing the reference's gate |A[r,c*]| > 1e-16, simplex.ts:370-375) x (columns of
 * the normalised pivot row that are non-zero, plus c*): the cells simplex.ts:376-387 reads and writes.
 */
typedef struct jslp_work_counters {
    int64_t relaxations;    /* restore + addCutConstraints + simplex units (relax / relax_batch / relax_from nodes)   */
    int64_t simplex_calls;  /* all simplex() runs, including plain jslp_engine_simplex                                */
    int64_t pivots;
    int64_t gated_cells;
    int64_t gated_rows;     /* sum over pivots of the rows passing the gate                                          */
    int64_t restored_rows;  /* rows copied back from the saved root / a checkpoint by restore()                      */
    int64_t cut_rows;       /* rows appended by addCutConstraints                                                    */
    int64_t height_sum;     /* sum over simplex calls of the tableau height (selection traffic = 8 x (W + 2H) per pivot) */
    /* health of the register-resident kernels (counted whether or not counting is on; set_counting resets them too): a solve whose
     * cooperative launch ended in a timed-out hand-off is rolled back and re-run through the streaming kernels -- the answer is
     * still the reference's, but a non-zero count means the fast path failed and nobody would notice from the result alone */
    int64_t resident_aborts;     /* register-resident launches rolled back (ERR_BARRIER inside the kernel)                  */
    int64_t resident_handovers;  /* solves the lean resident kernel handed on mid-solve (cycle-check history beyond its room) */
    int64_t resident_launches;   /* cooperative launches of k_simplex_resident that were accepted                           */
    int64_t resident_refusals;   /* ... that the runtime refused (not co-resident): the solve took the streaming kernels     */
    int64_t node_queue_launches; /* node batches evaluated by ONE launch of resident workgroups pulling nodes from a queue   */
    int64_t resident_fetch_retries; /* lean resident kernels: looks at a pivot row that had to be repeated because the row's checksum
                                     * did not match its flag word yet (the end-to-end check of the cross-XCD hand-over: DESIGN.md 4) */
} jslp_work_counters;
int jslp_engine_set_counting(jslp_engine* e, int enabled); /* also resets the counters */
int jslp_engine_get_counters(jslp_engine* e, jslp_work_counters* out);

/*
 * Device pool: branch-and-bound shards at node granularity -- every relaxation is a pure function of the
 * saved root and its cut list (branch-and-cut.ts:33-37) -- so a pool is `primary` (the engine behind the host's Tableau)
 * plus one engine per further entry of devices[], each holding a copy of the primary's saved root.  devices[0] must be the
 * primary's device; an ordinal may repeat ("virtual devices": several engines, streams and host threads on one GPU).
 * sync_root fans the saved root out to every member (hipMemcpyPeerAsync over xGMI: the snapshot, its index maps, the
 * unrestricted / integer flags and the optional objectives; 6.9 - 22.6 MB for BASELINE.json's configs); relax_batch does
 * it by itself when the primary's root changed since the last fan-out.  relax_batch = jslp_engine_relax_batch with the
 * nodes split into one contiguous range per member, every member driven by its own host thread and stream, outcomes
 * landing in ONE pinned buffer (the _pinned variant hands that buffer out, valid until the next call on the pool).
 * The caller stays single-threaded and synchronous: the threads live inside the pool.  The primary's
 * live tableau is left holding the last node of ITS range.  destroy() frees the members it created, never the primary.
 */
typedef struct jslp_pool jslp_pool;
int jslp_pool_create(jslp_pool** out, jslp_engine* primary, const int32_t* devices, int32_t n_devices);
void jslp_pool_destroy(jslp_pool* p);
int jslp_pool_size(const jslp_pool* p);
int jslp_pool_sync_root(jslp_pool* p);
int jslp_pool_relax_batch(jslp_pool* p, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                          const int32_t* var_index, const double* value, int check_cycles, jslp_simplex_result* out,
                          double* rhs, int32_t* var_index_by_row, int32_t out_stride);
int jslp_pool_relax_batch_pinned(jslp_pool* p, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                                 const int32_t* var_index, const double* value, int check_cycles,
                                 jslp_simplex_result* out, const double** rhs, const int32_t** var_index_by_row,
                                 int32_t* out_stride);
/* Counters of all members added up (see jslp_work_counters); set_counting applies to every member. */
/* The compact read-back over the pool (round 4): what a branch-and-bound host reads between relaxations is rowByVarIndex and the RHS
 * cell of the INTEGER variables (mip-utils.ts:43-61, 100-126), not the whole column -- n_watched x 12 bytes per node instead of
 * height x 12.  set_watched_variables = jslp_engine_set_watched_variables on every member; relax_batch_watched[_pinned] =
 * jslp_engine_relax_batch_watched[_pinned] with the nodes split over the members like jslp_pool_relax_batch, every member's
 * outcomes landing in ONE pinned buffer [n_nodes x n_watched] (the _pinned variant hands it out, valid until the next call). */
int jslp_pool_set_watched_variables(jslp_pool* p, const int32_t* var_indexes, int32_t n);
int jslp_pool_relax_batch_watched(jslp_pool* p, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                                  const int32_t* var_index, const double* value, int check_cycles, jslp_simplex_result* out,
                                  int32_t* watched_row, double* watched_value);
int jslp_pool_relax_batch_watched_pinned(jslp_pool* p, int32_t n_nodes, const int32_t* cut_offsets, const int8_t* type,
                                         const int32_t* var_index, const double* value, int check_cycles, jslp_simplex_result* out,
                                         const int32_t** watched_row, const double** watched_value);
/* the watched-variable count of the pool (= every member's, or -1 when they differ: set them with jslp_pool_set_watched_variables) */
int32_t jslp_pool_watched_count(const jslp_pool* p);
int jslp_pool_set_counting(jslp_pool* p, int enabled);
int jslp_pool_get_counters(jslp_pool* p, jslp_work_counters* out);

/* Current dimensions (height grows with cuts, restore() puts it back). */
int jslp_engine_dims(const jslp_engine* e, int32_t* height, int32_t* width, int32_t* n_var_indexes);

/* RHS column + varIndexByRow of the live tableau (height entries each; either may be NULL). */
int jslp_engine_read_rhs(jslp_engine* e, double* rhs, int32_t* var_index_by_row);

/*
 * Full read-back for `Solve(model, precision, full=true)` and the post-solve editing API
 * (main.ts:142-144): matrix in the reference layout (height*width, stride width); maps may be NULL.
 * row_by_var_index / col_by_var_index have n_var_indexes entries (-1 = absent).
 */
int jslp_engine_download(jslp_engine* e, double* matrix, int32_t* var_index_by_row, int32_t* var_index_by_col,
                         int32_t* row_by_var_index, int32_t* col_by_var_index);

/*
 * Diagnostics used by the parity tests: the (row, col) arguments of every pivot() since the last upload
 * in order.  Returns the total count in *n_pivots and copies at most max_pairs pairs into row_col
 * (interleaved r0,c0,r1,c1,...).  The FNV-1a digest of SURVEY.md Appendix C is computed by the caller.
 * The trace holds 2^20 pivots; asking for the pairs of a longer run fails with JSLP_ERR_CAPACITY (the count is still
 * returned) instead of handing out a truncated trace.
 */
int jslp_engine_pivot_trace(jslp_engine* e, int32_t* row_col, int64_t max_pairs, int64_t* n_pivots);

/*
 * Which launch shape the last simplex() / relax() used: "workgroup" (one workgroup per tableau), "select+update"
 * (two chip-wide launches per pivot), "fused" (one launch per phase-2 pivot) or "resident" (the whole simplex() on a
 * register-resident tableau in one cooperative launch); "oracle" for the test library, "none" before the first solve.
 */
const char* jslp_engine_last_path(const jslp_engine* e);

/*
 * Measurement hooks (bench.py): device time in milliseconds and launch count of the dominant kernel
 * (the row-update stream of pivot()) accumulated since the last reset, measured with HIP events on the
 * engine's own stream.  Timing is off by default (it adds an event pair per launch).
 */
int jslp_engine_set_timing(jslp_engine* e, int enabled);
int jslp_engine_get_timing(jslp_engine* e, double* update_kernel_ms, int64_t* update_kernel_launches,
                           double* total_device_ms);

#ifdef __cplusplus
}
#endif
#endif /* JSLP_ENGINE_H */
