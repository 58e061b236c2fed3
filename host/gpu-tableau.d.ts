// Type declarations of host/gpu-tableau.js for the reference's TypeScript host (src/solver.ts, src/main.ts).
// The binding only needs structural access to the reference's own classes, so they are typed loosely here.

/** What `install` needs from the reference's `Tableau` class (src/tableau/tableau.ts). */
export interface TableauClass {
    prototype: Record<string, any>;
}

/** The reference's Solver instance (src/main.ts): needed for the service-selection seam and for early engine release. */
export interface SolverLike {
    Solve: (...args: any[]) => any;
    selectBranchAndCutService?: (model: any) => any;
    lastSolvedModel?: any;
}

export interface LoadOptions {
    /** path of addon/jslp_napi.node (default: ../addon/jslp_napi.node next to this file) */
    addon?: string;
    /** path of libjslp_hip.so (default: ../jslpsolver_amd/csrc/libjslp_hip.so); a missing library throws: there is no CPU fallback */
    library?: string;
}

export interface InstallOptions extends LoadOptions {
    /** the reference's SlackVariable class (src/expressions.ts): cut rows get real instances of it */
    SlackVariable?: new (id: string, index: number) => any;
    /**
     * the reference's solver instance: enables options.useIncremental over device checkpoints, `speculate`, and the release
     * of the engine as soon as a simplified-result Solve() returns
     */
    solver?: SolverLike;
    /** HIP device ordinal (default 0) */
    device?: number;
    /**
     * > 1: default B&B policy with n-node speculative batches (in-order commit: same results and relaxation counts as the
     * sequential walk); needs `solver`.  Default 16 when `solver` is given; 0 keeps the reference's one-node-at-a-time services.
     */
    speculate?: number;
    /**
     * tableaus with fewer cells (width x height) stay on the reference's own TypeScript path.  Default (measured,
     * profiles/r02_mincells_sweep.md): 8192 for LPs, 262144 for models with integer variables, 32768 for those when `speculate` > 1;
     * 0 sends everything to the engine.
     */
    minCells?: number;
    /**
     * HIP device ordinals of a device pool (jslp_pool_*): speculative batches are split over one engine per ordinal (the
     * first must be `device`); the same ordinal may appear several times ("virtual devices" on one GPU)
     */
    devices?: number[];
}

/** host part of a device-resident checkpoint (StateCheckpoint, src/tableau/incremental-branch-and-cut.ts:31-44) */
export interface Checkpoint {
    id: number;
    height: number;
    nVars: number;
    lastElementIndex: number;
    availableIndexes: number[];
    evaluation: number;
    feasible: boolean;
}

export interface BranchCut {
    type: "min" | "max";
    varIndex: number;
    value: number;
}

export interface RelaxationOutcome {
    res: {
        feasible: boolean; bounded: boolean; optimal: boolean; unboundedVarIndex: number; pivotsPhase1: number;
        pivotsPhase2: number; cyclePhase: number; cycleStart: number; cycleLength: number; height: number;
        objCell: number; evaluation: number;
    };
    rhs: Float64Array;
    rows: Int32Array;
}

/** dlopen the engine library through the addon; returns the backend name ("hip-gfx950") */
export function loadEngine(options?: LoadOptions): string;
/** override the hot-path methods of the reference's Tableau class; returns the function that undoes it */
export function install(Tableau: TableauClass, options?: InstallOptions): () => void;
/** full read-back (matrix + index maps) for Solve(model, precision, full = true) consumers */
export function sync<T>(tableau: T): T;
/** (row, col) of every pivot since the upload, interleaved; null when the tableau is not on the engine */
export function pivotTrace(tableau: any): Int32Array | null;
/**
 * destroy the tableau's engine now (its resources go to the library's pool).  The tableau keeps its last read-back; a later
 * simplex() / applyCuts() on it throws instead of silently running on stale data.
 */
export function release(tableau: any): void;
/**
 * bring the live tableau back into the JS object and detach it from the engine: what the wrappers of the reference's
 * post-solve editing API (addConstraint, removeVariable, ... src/tableau/dynamic-modification.ts) do before they run
 */
export function bringHome(tableau: any): void;
/** keep options.useIncremental solves entirely on the reference's CPU path (for hosts that install() without `solver`) */
export function guardIncremental(solver: SolverLike): () => void;
export function isOnEngine(tableau: any): boolean;
export function createCheckpoint(tableau: any): Checkpoint;
export function relaxFromCheckpoint<T>(tableau: T, checkpoint: Checkpoint, cuts: BranchCut[]): T;
export function releaseCheckpoint(tableau: any, checkpoint: Checkpoint): void;
export function relaxBatch(tableau: any, cutLists: BranchCut[][]): RelaxationOutcome[];
/**
 * relaxBatch with the compact read-back (jslp_engine_relax_batch_watched): per node only rowByVarIndex (-1 = not basic) and the
 * RHS cell of `varIndexes` (default: the model's integer variables)
 */
export function relaxBatchWatched(tableau: any, cutLists: BranchCut[][], varIndexes?: ArrayLike<number>):
    Array<{ res: RelaxationOutcome["res"]; rows: Int32Array; values: Float64Array }>;
export function commitOutcome<T>(tableau: T, cuts: BranchCut[], outcome: RelaxationOutcome): T;
export function backend(): string | null;
