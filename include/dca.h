/*
 * dca.h — C ABI of libdca_hip.so, the MI355X (gfx950) batched weighted-A* (BWAS)
 * node-expansion engine for DeepCubeA-style search.
 *
 * This header is the drop-in boundary.  Every entry point names the
 * reference interface it replaces (paths relative to forestagostinelli/DeepCubeA):
 *
 *   environments/environment_abstract.py:23-163   Environment API (Python, in-process)
 *   environments/cube3.py:48-171                  Cube3.next_state/prev_state/expand/is_solved/state_to_nnet_input
 *   environments/n_puzzle.py:46-231               NPuzzle.*  (same set)
 *   utils/pytorch_models.py:49-52                 F.one_hot(x.long(), depth).float().view(-1, D*depth)
 *   cpp/environments.cpp:92-126,222-256           C++ twins (PuzzleN / Cube3 getNextState, isSolved)
 *   cpp/parallel_weighted_astar.cpp:138-346       the native BWAS loop (argv/stdout/socket boundary, replaced)
 *   search_methods/astar.py:50-90,232-340         Instance / AStar (python BWAS semantics)
 *
 * Conventions
 *   - plain C, no torch / STL types.  All `const uint8_t*` / `void*` buffers are
 *     caller-owned DEVICE memory unless the parameter comment says "host".
 *   - every call takes a hipStream_t (passed as void*) and is stream-ordered and
 *     asynchronous unless documented otherwise; the library never frees or
 *     reallocates caller memory.
 *   - return value: 0 = ok, >0 = hipError_t, <0 = library error (DCA_E_*);
 *     dca_last_error() returns a thread-local message for the last failure.
 *   - state rows are row-major uint8: cube3 [n,54] sticker ids 0..53; puzzles
 *     [n,dim*dim] tile ids 0..dim*dim-1 (0 = blank) — the reference's own numpy layout.
 */
#ifndef DCA_H_
#define DCA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCA_ABI_VERSION 5

/* library error codes (negative; positive values are hipError_t) */
#define DCA_E_BADARG (-1)
#define DCA_E_NOMEM (-2)
#define DCA_E_CAPACITY (-3) /* node pool / closed table / OPEN exhausted */
#define DCA_E_STATE (-4)    /* call sequence violated (e.g. commit before pop_expand) */
#define DCA_E_NOTFOUND (-5)

/* environments (utils/env_utils.py:6-28 registry names) */
#define DCA_ENV_CUBE3 0   /* "cube3": 54 stickers, 12 moves */
#define DCA_ENV_NPUZZLE 1 /* "puzzle15/24/35/48": dim 4..7, 4 moves U D L R */
#define DCA_ENV_LIGHTSOUT 2 /* "lightsout7": dim 7, 49 cells in {0,1}, 49 moves (press cell a: it and its 4-neighbours flip) */
#define DCA_ENV_CUBE4 3   /* the 4x4x4 cube of the reference's C++ core (cpp/environments.cpp:263-370): 96 stickers, 24 moves
                           * (12 outer-layer turns, then 12 inner-slice turns; move a^1 is the inverse of move a); solved =
                           * every face shows one colour (sticker / 16).  Environment kernels only: the reference's Python
                           * harness has no Cube4 (astar.py:473-486), so there is no network input / search driver for it. */

/* one-hot element types for the fused encoder (utils/pytorch_models.py:49-52 emits f32) */
#define DCA_DT_F32 0
#define DCA_DT_F16 1
#define DCA_DT_BF16 2
#define DCA_DT_F16X3 3 /* dca_l1_onehot_gemm only: the f16x3 split operand (vh, vl, vh) per element, [m, 3*n_pad] fp16 */
#define DCA_DT_F16_PLANES 4 /* dca_l1_onehot_gemm only: dca_f16x3_gemm's operand — [m, n_pad] fp16 high halves, then [m, n_pad] low halves */
#define DCA_DT_E4M3 5 /* dca_l1_onehot_gemm only: OCP fp8 e4m3 bytes, saturating (dca_gemm8's operand; the caller folds the activation
                         scale into the layer's weights and bias) */

/* BWAS semantics: which reference implementation is reproduced */
#define DCA_SEM_PY 0  /* search_methods/astar.py: f64 cost, FIFO ties, CLOSED starts empty */
#define DCA_SEM_CPP 1 /* cpp/parallel_weighted_astar.cpp: f32 cost, root in CLOSED, deferred stop.  PARITY UNPINNED beyond
                         * four recorded answers of the reference binary (the binary needs boost and
                         * cannot be rebuilt here): moves, path cost, iterations and nodes generated match them; |OPEN| /
                         * |CLOSED| may differ < 1 % where equal float32 costs tie (std::priority_queue's tie order is
                         * unspecified; this library breaks ties by push order).  DCA_SEM_PY is the exact one. */

/* built-in deterministic heuristics (test / engine-only benchmarking) */
#define DCA_HEUR_MOD97 0  /* f32( ((sum_i s_i*(7i+3)) mod 97) ) / 50f                        */
#define DCA_HEUR_KNUTH3 1 /* f32( ((sum_i s_i*(7i+3)) * 2654435761 mod 2^32) / 2^32 * 3 )    */
#define DCA_HEUR_HASHU01 2 /* f32( 10 + 5 * (hash64(s) >> 11) / 2^53 )                        */
#define DCA_HEUR_ZERO 3   /* 0 (uniform-cost search; nnet_utils.py:271-272 all_zeros server) */
#define DCA_HEUR_MANHATTAN 4 /* puzzles: sum over tiles of |row-row*|+|col-col*| (admissible, consistent); cube3: 0 */

int dca_abi_version(void);
const char* dca_last_error(void);

/* ---- move tables (host pointers, static storage) -------------------------------------- */
/* cube3 gather map: next[i] = cur[perm[a*54+i]]  (≡ cube3.py:183-256 / environments.h:75-105) */
const uint8_t* dca_cube3_perm_table(void);                 /* [12*54] */
/* puzzle blank-swap table [dim*dim*4] (≡ n_puzzle.py:174-214 / environments.cpp:4-46) */
int dca_npuzzle_swap_table(int dim, uint8_t* out /*host [dim*dim*4]*/);

/* ---- a1,a2: single-action move (Cube3._move_np cube3.py:163-171; prev = action^1) ------ */
int dca_cube3_next_state(const uint8_t* in /*[n,54]*/, int64_t n, int action, uint8_t* out /*[n,54]*/, void* stream);
int dca_cube3_prev_state(const uint8_t* in, int64_t n, int action, uint8_t* out, void* stream);
/* a7: NPuzzle._move_np (n_puzzle.py:216-231); blank index recomputed per state like n_puzzle.py:51-53 */
int dca_npuzzle_next_state(const uint8_t* in /*[n,dim*dim]*/, int64_t n, int dim, int action, uint8_t* out, void* stream);
int dca_npuzzle_prev_state(const uint8_t* in, int64_t n, int dim, int action, uint8_t* out, void* stream);

/* LightsOut._move_np (lights_out.py:155-166) / LightsOut::getNextState (environments.cpp:168-180): pressing cell `action`
 * flips it and its in-board 4-neighbours (move matrix lights_out.py:33-44 / environments.cpp:133-154).  Every move is
 * its own inverse: prev_state == next_state (lights_out.py:52-53).  dim 7 ("lightsout7", the only size the reference's
 * C++ core and train.sh use). */
int dca_lightsout_next_state(const uint8_t* in /*[n,dim*dim]*/, int64_t n, int dim, int action, uint8_t* out, void* stream);

/* ---- a3-a6,a9: fused expansion -----------------------------------------------------------
 * One launch: all children of every parent (Cube3.expand cube3.py:129-161), plus — each
 * optional, pass NULL to skip — the network input colour index (state_to_nnet_input
 * cube3.py:77-85: sticker//9), its one-hot encoding (pytorch_models.py:49-52; row = pos*6+colour),
 * is_solved (cube3.py:71-75) and the 64-bit state hash defined below.
 * Output order everywhere: child index = parent*num_moves + move.                           */
int dca_cube3_expand_fused(const uint8_t* parents /*[n,54]*/, int64_t n,
                           uint8_t* children /*[n,12,54] or NULL*/,
                           uint8_t* color_idx /*[n*12,54] or NULL*/,
                           void* onehot /*[n*12,324] of onehot_dtype or NULL*/, int onehot_dtype,
                           uint8_t* is_solved /*[n*12] or NULL*/,
                           uint64_t* hash /*[n*12] or NULL*/, void* stream);
/* puzzles: nnet input = the tiles themselves (n_puzzle.py:84-89); one-hot depth dim*dim */
int dca_npuzzle_expand_fused(const uint8_t* parents /*[n,D]*/, int64_t n, int dim,
                             uint8_t* children /*[n,4,D] or NULL*/,
                             void* onehot /*[n*4,D*D] or NULL*/, int onehot_dtype,
                             uint8_t* is_solved /*[n*4] or NULL*/,
                             uint64_t* hash /*[n*4] or NULL*/, void* stream);

/* lightsout: nnet input = the cells themselves (lights_out.py:70-75); one-hot depth 6 (get_nnet_model, lights_out.py:80) */
int dca_lightsout_expand_fused(const uint8_t* parents /*[n,D]*/, int64_t n, int dim,
                               uint8_t* children /*[n,D,D] or NULL*/,
                               void* onehot /*[n*D,D*6] or NULL*/, int onehot_dtype,
                               uint8_t* is_solved /*[n*D] or NULL*/,
                               uint64_t* hash /*[n*D] or NULL*/, void* stream);

/* cube4 — the remaining environment of the reference's C++ core (cpp/environments.cpp:263-370, tables environments.h:125-145):
 * 96 stickers (face * 16 + row * 4 + col), 24 moves; getNextState = a permutation gather (perm_table: next[i] =
 * cur[perm[a][i]], [24][96]), prev_state(a) = next_state(a ^ 1), isSolved = every face shows one colour (sticker / 16).
 * The reference's Python harness cannot drive Cube4 (astar.py:473-486 has no state_dim for it): no network input, no one-hot,
 * no engine instantiation here either — the environment kernels (the DCA_ENV_CUBE4 instantiation of the same tile code). */
const uint8_t* dca_cube4_perm_table(void);
int dca_cube4_next_state(const uint8_t* states /*[n,96]*/, int64_t n, int action, uint8_t* out, void* stream);
int dca_cube4_prev_state(const uint8_t* states, int64_t n, int action, uint8_t* out, void* stream);
int dca_cube4_expand_fused(const uint8_t* parents /*[n,96]*/, int64_t n, uint8_t* children /*[n,24,96] or NULL*/,
                           uint8_t* is_solved /*[n*24] or NULL*/, uint64_t* hash /*[n*24] or NULL*/, void* stream);

/* ---- stand-alone pieces of the same path (used by the Environment mirror) --------------- */
int dca_is_solved(int env, int dim, const uint8_t* states, int64_t n, uint8_t* out /*[n]*/, void* stream);
int dca_hash64(const uint8_t* states, int64_t n, int state_dim, uint64_t* out /*[n]*/, void* stream);
/* cube3: color_idx[n,54] = states//9 ; puzzles: copy */
int dca_nnet_input(int env, int dim, const uint8_t* states, int64_t n, uint8_t* out, void* stream);
/* generic one-hot of a [n,D] uint8 index array with depth `depth` -> [n, D*depth] */
int dca_onehot(const uint8_t* idx, int64_t n, int state_dim, int depth, void* out, int dtype, void* stream);
/* built-in deterministic heuristics on raw states -> f32 */
int dca_heuristic_builtin(int heur_id, const uint8_t* states, int64_t n, int state_dim, float* out, void* stream);

/* ---- (f)-1: approximate-value-iteration update step (ctg_approx/avi.py:129-159 do_update) -----------
 * generate_states (cube3.py:96-127 / n_puzzle.py:100-134): state i = goal after k_i ~ U{back_lo..back_hi}
 * uniformly random REVERSE moves; counter-based RNG keyed by (seed, index0 + i, step) so shards are
 * reproducible and independent.  out_num_back[n] / out_moves[n, moves_stride] (the forward move undone at
 * each step, -as taken-) are optional (NULL).                                                        */
int dca_generate_states(int env, int dim, int64_t n, int back_lo, int back_hi, uint64_t seed, int64_t index0,
                        uint8_t* out_states /*[n,D]*/, int32_t* out_num_back, int8_t* out_moves, int moves_stride,
                        void* stream);
/* search_utils.bellman (search_utils.py:16-32) after the children were expanded and evaluated:
 * ctg_backup[i] = solved_parent[i] ? 0 : min_a(1 + h[i*A+a]) (max(h,0) first when clip_zero),
 * argmin[i] = first a attaining it (gbfs.py:108 np.argmin).  Either output may be NULL.            */
int dca_bellman_backup(const float* h_children /*[n*A]*/, const uint8_t* solved_parent /*[n] or NULL*/, int64_t n,
                       int num_moves, int clip_zero, float* ctg_backup /*[n]*/, int32_t* argmin /*[n]*/, void* stream);

/* state hash (a9).  The reference's hash VALUES are process-randomised SipHash (cube3.py:17-21)
 * or unpinned boost::hash_range (parallel_weighted_astar.cpp:104-111); what is pinned is key
 * equality.  This library defines, for a D-byte state split in little-endian 8-byte words
 * w_0..w_{ceil(D/8)-1} (last word zero-padded):
 *     h = 0x9E3779B97F4A7C15 ^ (D * 0xD6E8FEB86659FD93)
 *     for k: h ^= w_k; h *= 0xFF51AFD7ED558CCD; h ^= h >> 32
 *     h ^= h >> 33; h *= 0xC4CEB9FE1A85EC53; h ^= h >> 33
 * The HIP kernels and the CPU oracle are bit-exact on it.                                    */

/* ---- a10-a13: device-resident BWAS engine ----------------------------------------------
 * Replaces cpp/parallel_weighted_astar.cpp (argv/stdout/socket) and the Instance/AStar
 * bookkeeping of search_methods/astar.py:50-90,232-340.  One engine = one search instance
 * on the current device.  OPEN, CLOSED and the node pool live in HBM; the heuristic is
 * supplied between the two halves of an iteration:
 *
 *     dca_engine_reset(e, root)
 *     dca_engine_root_commit(e, h_root)                        (PY semantics only)
 *     loop:  dca_engine_pop_expand(e, &nnet_in, &onehot, &m)  ->  h = heuristic(nnet_in[m])
 *            dca_engine_commit(e, h)
 *            dca_engine_status(e, &st);  stop when st.done
 *     dca_engine_solution(e, moves, &len, &path_cost)
 */
typedef struct dca_engine dca_engine;

typedef struct dca_status {
    int32_t done;            /* search finished (goal popped / cpp stop rule hit)            */
    int32_t failed;          /* capacity exhausted or OPEN ran empty                          */
    int64_t iterations;      /* completed pop/expand/commit iterations                        */
    int64_t nodes_generated; /* reference's "# Nodes Gen" (astar.py:168 / cpp:166,266)        */
    int64_t nodes_expanded;  /* parents popped and expanded                                   */
    int64_t open_size;
    int64_t closed_size;
    int64_t pool_size;       /* node ids handed out                                           */
    double best_cost;        /* cost of the cheapest solved node popped so far (cpp) / NaN    */
} dca_status;

/* onehot_dtype: DCA_DT_* to have pop_expand also emit the one-hot rows of the batch's children
 * (fused into the expansion launch), or -1 for none.  max_nodes bounds the node pool (one id per
 * generated child), the CLOSED table (2x, power of two) and OPEN.                                 */
int dca_engine_create(dca_engine** out, int env, int dim, double weight, int batch_size,
                      int64_t max_nodes, int semantics, int onehot_dtype);
/* K independent search instances of the same geometry in ONE engine: every kernel is launched once for all of
 * them (grid.y = instance), the way the reference's AStar steps a list of instances together
 * (astar.py:232-317).  A batch-20 000 iteration is launch/latency bound, so K scrambles advance in little more
 * than the time of one.  Instances are inert until reset with a root; each has its own OPEN/CLOSED/pool
 * (max_nodes each).  The un-suffixed calls below act on instance 0; pop_expand/commit/run_builtin act on all. */
int dca_engine_create_multi(dca_engine** out, int env, int dim, double weight, int batch_size,
                            int64_t max_nodes, int semantics, int onehot_dtype, int num_instances /*1..64*/);
int dca_engine_num_instances(dca_engine* e);
void dca_engine_destroy(dca_engine* e);
int dca_engine_reset(dca_engine* e, const uint8_t* root /*host [D]*/, void* stream);
int dca_engine_reset_instance(dca_engine* e, int inst, const uint8_t* root /*host [D]*/, void* stream);
/* PY semantics: cost(root) = w*0 + max(h,0)*!solved (astar.py:244-249). h_root: device f32[1].
 * CPP semantics never evaluates the root (cpp:160) — the call is then a no-op.                    */
int dca_engine_root_commit(dca_engine* e, const float* h_root, void* stream);
int dca_engine_root_commit_instance(dca_engine* e, int inst, const float* h_root, void* stream);
/* network-input row of the root (device [D]; cube3: colour index), valid after reset            */
int dca_engine_root_nnet_in(dca_engine* e, const uint8_t** nnet_in);
int dca_engine_root_nnet_in_instance(dca_engine* e, int inst, const uint8_t** nnet_in);
/* first half: pop min(B,|OPEN|) by (cost, push order), expand.  *nnet_in = device pointer to the
 * network-input rows [K*batch*num_moves, D] (instance-major) of the batch's children (cube3: colour index; puzzles:
 * the tiles), child index = pop_rank*num_moves + move; *onehot = their one-hot rows
 * [K*batch*num_moves, D*depth] (NULL unless enabled at create).  Both buffers have the FIXED row
 * count m_capacity = K*batch*num_moves so the heuristic can be enqueued without a host sync; rows
 * past the live child count hold stale but valid rows and their heuristic values are ignored.     */
int dca_engine_pop_expand(dca_engine* e, const uint8_t** nnet_in, const void** onehot,
                          int64_t* m_capacity, void* stream);
/* second half: h = device f32[m_capacity].  max(h,0) is applied here (nnet_utils.py:193-194
 * clip_zero=True): cost, CLOSED dedup, push.                                                      */
int dca_engine_commit(dca_engine* e, const float* h, void* stream);
/* Dedup-first ("packed") stepping.  The reference runs the network on every child and only then drops the
 * children already in CLOSED (astar.py:272-282 "do heur before check"); a dropped child's value is never used, so
 * checking first and evaluating only the survivors gives the identical search with ~15 % (cube3) to ~40 %
 * (sliding puzzles) fewer network rows, and none at all for the padding rows of a short batch.
 *   enable_packed    once after create: allocates the packed batch buffers.  onehot_dtype DCA_DT_* / -1;
 *                    onehot_row_stride >= D*depth elements with stride*sizeof(elt) a multiple of 16 (rows are
 *                    written with a zero tail so a GEMM can use the padded K directly).
 *   pop_expand_packed  pop, expand, CLOSED check, pack.  *rows = kept children of all instances this iteration
 *                    (synchronises to return it); *nnet_in [rows, D], *onehot [rows, stride], *src [rows] =
 *                    instance*batch*num_moves + child index of each packed row.  The buffers are sized to
 *                    K*batch*num_moves rounded up to 1024 rows; rows past *rows hold stale but finite data.
 *   commit_packed    h = device f32[rows] in packed row order: cost and push of the kept children.        */
int dca_engine_enable_packed(dca_engine* e, int onehot_dtype, int64_t onehot_row_stride);
int dca_engine_pop_expand_packed(dca_engine* e, const uint8_t** nnet_in, const void** onehot, const uint32_t** src,
                                 int64_t* rows, void* stream);
int dca_engine_commit_packed(dca_engine* e, const float* h, void* stream);
/* what the last pop_expand_packed found besides its row count (no extra sync: read with it): how many of the engine's
 * instances were already finished when it ran (done: goal popped and accepted, OPEN empty, or failed) and how many of those
 * failed (node pool exhausted, ...).  A host loop stops stepping when instances_done == K instead of polling the status. */
int dca_engine_packed_state(dca_engine* e, int* instances_done, int* instances_failed);
/* both halves with a built-in heuristic (evaluated inside the expansion launch), `iters`
 * iterations enqueued without any host sync (kernels no-op once the search is done).
 * use_graph != 0 replays one captured hipGraph per iteration instead of eager launches.           */
int dca_engine_run_builtin(dca_engine* e, int heur_id, int iters, int use_graph, void* stream);
/* synchronises the stream */
int dca_engine_status(dca_engine* e, dca_status* out, void* stream);
int dca_engine_status_instance(dca_engine* e, int inst, dca_status* out, void* stream);
/* ASTAR updates (updaters/updater.py:36-54 of the reference: one batch-1 search per training state, each with its own random
 * weight, stepped together; every popped node becomes a training target).
 *   set_weight_instance  weight of path cost of one instance (astar.py:196 `weights`), between iterations; set_weights: of the
 *                        first n instances at once (one synchronisation);
 *   park_instance        marks an instance finished (its launches become no-ops) until its next reset; between iterations
 *                        only (DCA_E_STATE between pop_expand and commit: the commit half records the CLOSED slots the
 *                        expansion half claimed);
 *   last_popped          between pop_expand and commit: the parents of this pop, instance-major — states device u8
 *                        [K*batch, D], flags device u8 [K*batch]: 0 = nothing popped in that slot, 1 = popped, 2 = popped and
 *                        solved (Node.is_solved: its backup is 0, astar.py:38-40). */
int dca_engine_set_weight_instance(dca_engine* e, int inst, double weight);
int dca_engine_set_weights(dca_engine* e, const double* weights /*host [n], instances 0..n-1*/, int n);
int dca_engine_park_instance(dca_engine* e, int inst, void* stream);
int dca_engine_last_popped(dca_engine* e, uint8_t* states, uint8_t* flags, void* stream);
/* The same controls without a host round trip, for stepping thousands of batch-1 searches (one ASTAR update = up to 5e5 of
 * them, updater.py:36-54): everything is enqueued on `stream`, nothing synchronises.
 *   reset_many        instances 0..n-1 restart from roots_dev (device u8 [n, D]; NOT range-checked — the rows come from the
 *                     library's own generator), instances n..K-1 are parked;
 *   root_commit_many  h_roots_dev: device float [n], the heuristic of the n roots;
 *   set_weights_dev   weights of path cost of instances 0..n-1 from a device double [n] (stream-ordered, nothing
 *                     synchronises); they hold until a host-side dca_engine_set_weights / dca_engine_set_weight_instance
 *                     names the same instance: every call that re-uploads the host's copy of the instance table (set_tiers,
 *                     the profile calls, the host setters) first reads the device's weights back into it.  A weight < 0 or
 *                     NaN — which the host setters refuse with DCA_E_BADARG — is clamped to 0 by the launch. */
int dca_engine_reset_many(dca_engine* e, const uint8_t* roots_dev, int n, void* stream);
int dca_engine_root_commit_many(dca_engine* e, const float* h_roots_dev, int n, void* stream);
int dca_engine_set_weights_dev(dca_engine* e, const double* weights_dev, int n, void* stream);
/* facts about an engine (host int64[8]): [0] workgroups of k_sel_collect's grid, [1] how many of them the device holds at once
 * according to hipOccupancyMaxActiveBlocksPerMultiprocessor (-1: query failed), [2] 1 if the grid-wide refinement of giant tie
 * bins (grid barriers; needs [1] >= [0]) was enabled at creation, [3] bytes of the CLOSED table, [4] 1 once a grid barrier
 * found the launch not fully resident (the GPU is shared) and the engine fell back to the single-workgroup path for good;
 * [5..7] reserved.  Synchronises.  (knob 10 of dca_debug_tune makes the first barrier of every giant iteration give up at
 * once: the test hook for that fallback.) */
int dca_engine_info(dca_engine* e, int64_t* out /*host [8]*/, void* stream);
/* child rows of the last pop_expand straight from the node pool (device [m_live, D]); synchronises */
int dca_engine_last_children(dca_engine* e, const uint8_t** states, int64_t* m_live, void* stream);
/* root->goal move list (astar.py:213-229 get_path / cpp:336-341).  synchronises.                 */
int dca_engine_solution(dca_engine* e, int32_t* moves /*host [cap]*/, int cap, int* len, double* path_cost, void* stream);
int dca_engine_solution_instance(dca_engine* e, int inst, int32_t* moves /*host [cap]*/, int cap, int* len,
                                 double* path_cost, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Training step: BatchNorm1d in training mode over row-major [n, c] float32 activations, as
 * nn.BatchNorm1d runs inside nnet_utils.train_nnet (utils/nnet_utils.py:53-118, utils/pytorch_models.py:57-86),
 * optionally fused with the residual add and the ReLU that follow it in the network:
 *   forward   y = relu?( (x - mean_c) * invstd_c * gamma_c + beta_c (+ skip) ), batch statistics over the n rows
 *             (biased variance normalises; var_unbiased feeds running_var, momentum is the caller's business)
 *   backward  g = dy * (y > 0 if relu); dbeta = sum g; dgamma = sum g*xhat;
 *             dx = gamma*invstd*(g - dbeta/n - xhat*dgamma/n); dskip = g (if dskip != NULL)
 * Column sums are accumulated in fp64 and folded deterministically.  workspace: dca_bn_workspace_bytes(c) bytes.
 * ------------------------------------------------------------------------------------------------------------------ */
int64_t dca_bn_workspace_bytes(int64_t c);
int dca_bn_train_forward(const float* x, const float* skip /*or NULL*/, const float* gamma, const float* beta, int64_t n,
                         int64_t c, double eps, int relu, float* y, float* mean /*[c]*/, float* invstd /*[c]*/,
                         float* var_unbiased /*[c]*/, void* workspace, int64_t workspace_bytes, void* stream);
int dca_bn_train_backward(const float* dy, const float* x, const float* y /*needed if relu*/, const float* mean,
                          const float* invstd, const float* gamma, int64_t n, int64_t c, int relu, float* dx,
                          float* dskip /*or NULL*/, float* dgamma /*[c]*/, float* dbeta /*[c]*/, void* workspace,
                          int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Heuristic network, first layer: y = relu?( onehot(states_nnet, depth) . W1^T + b1 ) straight from the
 * uint8 network-input rows — utils/pytorch_models.py:49-60 with BatchNorm folded — as one MFMA kernel that never
 * materialises the one-hot matrix.  Weights arrive as `planes` bf16 planes whose sum is the fp32 weight matrix
 * (planes = 3: exact fp32 products => an fp32 GEMM on the bf16 MFMA pipes; 2 for fp16 weights; 1 for bf16), tiled as
 * [n_pad/64][planes][k_pad/8][64][8] (k_pad = dca_l1_kpad(state_dim, depth), zero padded; deepcubea_amd/utils/
 * pytorch_models.py:l1_weight_tiles builds it).  out: [m, n_pad] in out_dtype (DCA_DT_*), row stride n_pad; DCA_DT_F16X3
 * writes the next f16x3 layer's A operand [m, 3*n_pad] directly (see dca_act_split).
 * K is walked in LDS-sized chunks (cube3: one piece; the sliding puzzles, K up to 2401: 320 one-hot columns at a time).
 * Instantiated for cube3 (54, 6) and the sliding puzzles (16/25/36/49): dca_l1_supported(state_dim, depth) != 0.
 * ------------------------------------------------------------------------------------------------------------------ */
int dca_l1_supported(int state_dim, int depth);
int64_t dca_l1_kpad(int state_dim, int depth);
int dca_l1_onehot_gemm(const uint8_t* nnet_in /*[m, state_dim]*/, int64_t m, int state_dim, int depth, const void* w_tiles,
                       int planes, int64_t n_pad, const float* bias /*[n_pad]*/, int relu, void* out, int out_dtype,
                       int* overflow /*device flag, set to 1 if a DCA_DT_F16X3 value exceeds fp16; or NULL*/, void* stream);

/* The same layer in the NON-parity fp8 mode, on gfx950's f8f6f4 matrix pipe (csrc/dca_mlp8.hip): a one-hot row is exact in
 * OCP e4m3, so with the layer's weights as e4m3 bytes + one fp32 scale per output unit (what every other fp8 layer carries)
 *   out8[r, n] = e4m3(sat( relu?( (onehot(s_r) . w8[n, :]) * scale[n] + bias[n] ) ))
 * runs on v_mfma_f32_32x32x64_f8f6f4 — twice the rate of the bf16 pipe dca_l1_onehot_gemm(..., DCA_DT_E4M3) multiplies the same
 * exact 0 / 1 rows on.  The caller folds the activation scale of the output tensor into scale[] and bias[].
 * w_tiles: [n_pad/128][k_pad/16][128][16] e4m3 bytes, k_pad = dca_l1_kpad8(state_dim, depth) (K zero padded to a multiple of 64;
 * deepcubea_amd/utils/pytorch_models.py:l1_weight_tiles8 builds it); n_pad % 128 == 0; nnet_in and out8 16-byte aligned;
 * out8: [m, n_pad] bytes.  Instantiated where the 128-column weight tile fits LDS: dca_l1_supported8() != 0 (cube3,
 * puzzle15, puzzle24, lightsout7); other geometries keep the bf16-pipe kernel. */
int dca_l1_supported8(int state_dim, int depth);
int64_t dca_l1_kpad8(int state_dim, int depth);
int dca_l1_onehot_gemm8(const uint8_t* nnet_in /*[m, state_dim]*/, int64_t m, int state_dim, int depth, const void* w_tiles,
                        int64_t n_pad, const float* scale /*[n_pad]*/, const float* bias /*[n_pad]*/, int relu, void* out8,
                        void* stream);

/* The same layer as an EMBEDDING SUM on the vector pipes (csrc/dca_embed.hip): a one-hot row has one 1 per position, so
 *   out[r, n] = relu?( bias[n] + sum_pos w_t[pos * depth + s[r, pos]][n] )          (positions in ascending order, fp32 adds)
 * — state_dim gathered weights per output instead of state_dim * depth multiply-adds, in exact fp32 arithmetic with no operand
 * splitting.  It pays where depth is large (the sliding puzzles: depth = the tile count; puzzle48: 49 of 2401 columns are ones);
 * cube3 (depth 6) is faster on the matrix pipes (dca_l1_onehot_gemm).  w_t: the layer's weight matrix TRANSPOSED, fp32
 * [state_dim * depth][n_pad] (row = one-hot column); bias [n_pad]; n_pad % 64 == 0; nnet_in, w_t, out 16-byte aligned.
 * out_dtype: DCA_DT_F32 [m, n_pad], DCA_DT_BF16 [m, n_pad], or DCA_DT_F16_PLANES (dca_f16x3_gemm's operand: [m, n_pad] fp16
 * high halves, then [m, n_pad] low halves; *overflow set to 1 if a value exceeds fp16), or DCA_DT_E4M3 [m, n_pad] bytes,
 * saturating (the caller has folded the activation scale into w_t and bias).  State bytes must be < depth.
 * Instantiated for the geometries dca_l1_embed_supported() names (cube3, puzzle15/24/35/48, lightsout7). */
int dca_l1_embed_supported(int state_dim, int depth);
int dca_l1_embed(const uint8_t* nnet_in /*[m, state_dim]*/, int64_t m, int state_dim, int depth, const float* w_t, int64_t n_pad,
                 const float* bias /*[n_pad]*/, int relu, void* out, int out_dtype, int* overflow /*device flag or NULL*/,
                 void* stream);

/* Glue of the fp32-accurate "f16x3" dense layers (csrc/dca_mlp.hip): v = relu?(y*alpha*col_scale + bias (+ skip)) over the row-major
 * fp32 GEMM output y [m, n]; writes the next layer's A operand a3 [m, 3n] fp16, a3[3k..3k+2] = (vh, vl, vh) with
 * vh = f16(v), vl = f16(v - vh), and, if x_out != NULL, v itself (the next residual block's skip).  With the weights as
 * W3[3k..3k+2] = (wh, wh, wl) (row n pre-scaled by the power of two 1/(alpha*col_scale[n])) one f16 GEMM with fp32 output reproduces the fp32 layer
 * utils/pytorch_models.py:57-86 computes (BatchNorm folded) to fp32 accuracy.  n % 4 == 0.                            */
int dca_act_split(const float* y, const float* bias /*[n] or NULL*/, const float* skip /*[m,n] or NULL*/,
                  const float* col_scale /*[n] or NULL: per-output-unit inverse weight scale*/, double alpha, int relu,
                  int64_t m, int64_t n, float* x_out /*[m,n] or NULL*/, void* a3 /*[m,3n] fp16 or NULL*/,
                  int a3_planes /*!= 0: write dca_f16x3_gemm's operand instead: [m,n] high halves then [m,n] low halves*/,
                  int* overflow /*device flag, set to 1 if some |v| > 60000 (not splittable into fp16); or NULL*/,
                  void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Heuristic network, dense layers: v = relu?( (x . W^T) * alpha * col_scale + bias (+ skip) ) — one layer
 * of utils/pytorch_models.py:57-86 with BatchNorm folded — as ONE hand-written MFMA kernel with fp32 accuracy on the f16
 * matrix pipes ("f16x3": x = xh + xl, w = wh + wl in fp16; x.w = xl.wh + xh.wl + xh.wh accumulated in fp32; csrc/dca_gemm.hip).
 * Operands are fp16 PLANES: a_h / a_l [m, lda] (high / low halves of the fp32 activations: what this kernel, dca_l1_onehot_gemm
 * (DCA_DT_F16_PLANES), dca_act_split (a3_planes) and dca_split_planes emit), w_h / w_l [n, ldw] (rows = output units,
 * pre-scaled by the power of two 1 / col_scale[row]).  k % 64 == 0; lda, ldw % 8 == 0; 16-byte aligned bases.
 * Outputs (row stride ldo): out_h / out_l — the result's planes, i.e. the NEXT layer's operand — and / or x_out, the fp32
 * result (the next residual block's skip, or the input of the 1-wide output layer).  overflow: device flag set when a value
 * does not fit fp16 (|v| > 60000; the caller then redoes the batch with fp32 GEMMs), or NULL.
 * ------------------------------------------------------------------------------------------------------------------ */
int dca_f16x3_gemm(const void* a_h, const void* a_l, int64_t m, int k, int64_t lda, const void* w_h, const void* w_l, int n,
                   int64_t ldw, const float* col_scale /*[n] or NULL*/, double alpha, const float* bias /*[n] or NULL*/,
                   const float* skip /*[m, ldo] fp32 or NULL*/, int relu, void* out_h, void* out_l, float* x_out, int64_t ldo,
                   int* overflow, void* stream);

/* The same layer in the NON-parity 16-bit modes (`--nnet_dtype bf16 | fp16`; replaces the library GEMM + separate clamp pass of
 * utils/pytorch_models.py:57-86 as PyTorch runs it): out = relu?( a . w^T + bias (+ skip) ), operands and result in `dtype`
 * (DCA_DT_BF16 or DCA_DT_F16), fp32 accumulation on the 16-bit MFMA pipes, bias fp32, the residual `skip` [m, ldo] in
 * `dtype`, result rounded to nearest even — bias, residual add, ReLU and rounding ride in the epilogue (csrc/dca_gemm16.hip).
 * a [m, lda], w [n, ldw] (rows = output units); k % 64 == 0; lda, ldw % 8 == 0; ldo % 4 == 0; a / w 16-byte, out / skip
 * 8-byte aligned.  `out` may not alias `a`; it may alias `skip`. */
int dca_gemm16(const void* a, int64_t m, int k, int64_t lda, const void* w, int n, int64_t ldw, int dtype,
               const float* bias /*[n] or NULL*/, const void* skip /*[m, ldo] or NULL*/, int relu, void* out, int64_t ldo,
               void* stream);

/* The same layer in the NON-parity fp8 mode (`--nnet_dtype fp8`; csrc/dca_gemm8.hip): OCP e4m3 operands a [m, lda] / w [n, ldw]
 * (bytes; k % 128 == 0, lda / ldw % 16 == 0), fp32 accumulation on v_mfma_f32_32x32x64_f8f6f4, and the whole tail in the epilogue:
 *   v = relu?( (a . w^T)[m,n] * scale[n] + bias[n] (+ skip[m,n]) )        scale[n] = activation scale x weight scale of unit n
 * leaving as bf16 (out16 [m, ldo16], the residual stream; may be `skip` itself) and / or quantised for the next layer:
 * out8 [m, ldo8] = e4m3(sat(v * out8_scale)).  skip is bf16 with out16's row stride.  Replaces utils/pytorch_models.py:57-86
 * (BatchNorm folded) as PyTorch runs it, at fp8 operand precision: never a parity mode. */
int dca_gemm8(const void* a, int64_t m, int k, int64_t lda, const void* w, int n, int64_t ldw, const float* scale /*[n]*/,
              const float* bias /*[n] or NULL*/, const void* skip /*bf16 [m, ldo16] or NULL*/, int relu, void* out16 /*or NULL*/,
              int64_t ldo16, void* out8 /*or NULL*/, int64_t ldo8, double out8_scale, void* stream);
/* Block-scaled ("MX") form of the fp8 layer — what `--nnet_dtype fp8` runs: activations travel as e4m3 bytes PLUS one E8M0 scale
 * byte (value 2^(byte - 127)) per row and 64 consecutive elements, computed where the activation is produced (the largest
 * magnitude of the 64 values sets the power of two that brings them into e4m3's range) and applied by gfx950's scaled MFMA
 * (v_mfma_scale_f32_32x32x64_f8f6f4: both lane halves of a row pass the row's scale of the 64-deep step; the weights pass
 * 2^0 and keep their per-output-unit fp32 scales w_scale[n] for the epilogue).  No calibration, nothing frozen, no saturation:
 *   v = relu?( (sum_blocks 2^(sa - 127) * a8 . w8^T)[m,n] * w_scale[n] + bias[n] (+ skip[m,n]) )
 * a_scale [m, ld_asc] (k / 64 bytes per row; k % 256 == 0, k <= 8192); out8 / out8_scale [m, ld_osc] (n / 64 bytes per row,
 * n % 64 == 0) = the next layer's operand and its block scales, or both NULL; out16 / skip as in dca_gemm8.
 * dca_l1_onehot_gemm_mx: layer 1 (one bf16 weight plane, dca_l1_onehot_gemm's tiles) leaving in the same form. */
int dca_gemm8_mx(const void* a, const void* a_scale, int64_t m, int k, int64_t lda, int64_t ld_asc, const void* w, int n,
                 int64_t ldw, const float* w_scale /*[n]*/, const float* bias /*[n] or NULL*/,
                 const void* skip /*bf16 [m, ldo16] or NULL*/, int relu, void* out16 /*or NULL*/, int64_t ldo16,
                 void* out8 /*or NULL*/, int64_t ldo8, void* out8_scale /*or NULL*/, int64_t ld_osc, void* stream);
int dca_l1_onehot_gemm_mx(const uint8_t* nnet_in /*[m, state_dim]*/, int64_t m, int state_dim, int depth, const void* w_tiles,
                          int64_t n_pad, const float* bias /*[n_pad]*/, int relu, void* out8 /*[m, n_pad] e4m3*/,
                          void* out_scale /*[m, ld_sc] E8M0*/, int64_t ld_sc, void* stream);
/* x [m, n] (row stride ld; DCA_DT_F32 or DCA_DT_BF16) -> e4m3(sat(x * scale)) bytes [m, ldo]: the entry into an fp8 layer for
 * activations that did not come out of an fp8 epilogue.  n % 4 == 0. */
int dca_quant_e4m3(const void* x, int dtype, int64_t m, int64_t n, int64_t ld, double scale, void* out, int64_t ldo, void* stream);
/* fp32 [m, n] (row stride ld) -> its fp16 planes (row stride ldo); n % 4 == 0 */
int dca_split_planes(const float* x, int64_t m, int64_t n, int64_t ld, void* out_h, void* out_l, int64_t ldo,
                     int* overflow /*or NULL*/, void* stream);

/* Operands of the TRAINING step's dense layers (nn.Linear forward y = x . W^T and backward dx = dy . W of
 * utils/nnet_utils.py:53-118 through dca_f16x3_gemm; deepcubea_amd/_lib.py linear_train).  Nothing is prepared ahead here —
 * the weights change every step, gradients span many binades — so every operand is scaled by a power of two taken from its own
 * magnitude before it is split (exact), and the inverse scales return through the GEMM's col_scale:
 *   dca_absmax_bits          max |x| of an fp32 matrix as float bits, into a device word (zeroed here)
 *   dca_split_planes_scaled  x * 2^e -> fp16 planes, 2^e taking max|x| (amax_bits, from dca_absmax_bits) into [2^14, 2^15);
 *                            amax_bits NULL = unscaled; columns n..n_pad are written as zeros (the GEMM wants k % 64 == 0)
 *   dca_split_rows_scaled    per ROW of w [n, k]: the row's own 2^e, its planes (columns k..k_pad zero) and
 *                            col_scale[row] = 1 / (2^e * the other operand's scale, when other_amax_bits is given)
 * n, k, n_pad, k_pad % 4 == 0; 16-byte aligned inputs. */
int dca_absmax_bits(const float* x, int64_t m, int64_t n, int64_t ld, uint32_t* out_bits, void* stream);
int dca_split_planes_scaled(const float* x, int64_t m, int64_t n, int64_t ld, const uint32_t* amax_bits /*device or NULL*/,
                            void* out_h, void* out_l, int64_t ldo, int64_t n_pad, void* stream);
int dca_split_rows_scaled(const float* w, int64_t n, int64_t k, int64_t ld, void* out_h, void* out_l, int64_t ldo, int64_t k_pad,
                          float* col_scale /*[n]*/, const uint32_t* other_amax_bits /*device or NULL*/, void* stream);
/* dca_fill_inv_pow2: out[0..n) = 2^-e of the power-of-two scale dca_split_planes_scaled derives from amax_bits. */
int dca_fill_inv_pow2(float* out, int64_t n, const uint32_t* amax_bits, void* stream);

/* Output layer of the cost-to-go network (utils/pytorch_models.py:83-86, fc_out: res_dim -> out_dim, out_dim = 1 for every
 * environment of the reference): out[m, n_out] = x[m, k] . w[n_out, k]^T + bias, float64 accumulation in a FIXED order (one wave
 * per row, lane-strided FMA chains, xor-butterfly fold, one rounding to fp32): a row's value does not depend on its position, on m or on the launch
 * — the library GEMV this replaces chose its kernel (and summation order) from m.  x: DCA_DT_F32 / F16 / BF16 rows (row
 * stride ldx elements, k % 4 == 0, ldx % 4 == 0); w, bias, out fp32; n_out <= 8. */
int dca_head_gemv(const void* x, int x_dtype, int64_t m, int k, int64_t ldx, const float* w /*[n_out, k]*/,
                  const float* bias /*[n_out] or NULL*/, int n_out, float* out /*[m, n_out]*/, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DCA_H_ */
