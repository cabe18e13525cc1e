/* dca_debug.h — test, tuning and profiling hooks of libdca_hip.so, kept apart from the product ABI (include/dca.h).
 *
 * Nothing here is needed to run a search, an update or a network forward: these entry points exist for tests/ (race screens
 * against a plain schedule, forced fallbacks, tiny tiers), for bench.py's device-side launch profile and for the A/B tools
 * under tools/.  Same conventions as dca.h (plain C, int return codes, the error text of the last failure through dca.h).  Exported by the same library. */
#ifndef DCA_DEBUG_H
#define DCA_DEBUG_H

#include "dca.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- engine ------------------------------------------------------------------------------------------------------ */
/* Host-only (no device needed): the chunk dca_engine_run_builtin would replay as ONE hipGraph for a run of `remaining`
 * iterations starting at iteration `host_iter` of a search — its length n (<= 64: up to the next rebase-period boundary, then
 * whole periods) and which of its iterations are rebase iterations (bit i of rebase_mask).  For tests of the cutting rule. */
int dca_engine_plan_chunk(int64_t host_iter, int remaining, int* n, uint64_t* rebase_mask);

/* Device-side profile of `iters` iterations of run_builtin (use_graph as there): every workgroup stamps the device wall
 * clock at entry and exit, per launch the host takes max(end) - min(start) as the launch's busy span and the distance to
 * the previous launch's end as the gap in front of it — measured INSIDE the replayed hipGraph, which HIP events between
 * eager launches cannot do.  span_ms / gap_ms: host float[DCA_PROF_SLOTS], summed milliseconds over the iterations
 * (gap_ms may be NULL).  Slots: 0 refill_hist 1 refill_scan 2 refill_move 3 sel_hist (= the FRONT rebase pass; 0-3 only in
 * rebase iterations: every 8th, and the first twelve after a reset) 4 sel_scan 5 sel_collect 6 rank 7 expand 8 probe 9 decide
 * 10 pack (dedup-first stepping only) 11 commit; 12 / 13 = the two halves of the rank launch (small-bin pass, large-bin
 * workgroups), 14-17 = phases of the large-bin path (load + range, count + prefix, scatter, order) as envelopes over the
 * workgroups — for tuning.  Synchronises every iteration.                                                            */
#define DCA_PROF_SLOTS 18
int dca_engine_profile_builtin(dca_engine* e, int heur_id, int iters, int use_graph, float* span_ms /*host [DCA_PROF_SLOTS]*/,
                               float* gap_ms /*host [DCA_PROF_SLOTS] or NULL*/, void* stream);

/* test / tuning hook: FRONT-tier hysteresis in entries (defaults 32*B / 96*B); results never depend on it */
int dca_engine_set_tiers(dca_engine* e, int64_t front_keep, int64_t front_max);

/* internals of the last iteration for diagnostics (host double[16]; layout in dca_engine.hip); synchronises */
int dca_engine_debug(dca_engine* e, double* out /*host [16]*/, void* stream);
/* diagnostics: flips a tuning knob of the engine kernels process-wide (0 = shipped behaviour); never needed in production.
 * knob 0: extra log2 of sub-bins per large bin in k_rank; 1: sub-bin size above which a sub-bin is refined on its own;
 * 2: BACK squeeze mark (1/1024ths of max_nodes); 3: threshold-bin size above which the grid refines the bin; 4: workgroups of
 * k_sel_collect, 5: grid-wide refinement off, 6: k_sel_scan in every iteration, 7: single-iteration graphs only (4-7 host side,
 * set before the engine is created / first stepped); 9: largest bin k_rank orders a thread per entry; 0-15 accepted.      */
int dca_debug_tune(int knob, int value);

/* ---- environment kernels ------------------------------------------------------------------------------------------- */
/* Store-only yardstick of the gather kernel's roofline (bench.py times it in the same process, right after the kernel itself):
 * fills buf[0, bytes) with plain 16-byte stores, one contiguous region of bytes_per_block per workgroup (1 MiB: the pattern
 * tools/hbm_write_ceiling.hip found best on an MI355X).  buf 16-byte aligned; bytes and bytes_per_block multiples of 16. */
int dca_debug_write_ceiling(void* buf, int64_t bytes, int64_t bytes_per_block, void* stream);

/* ---- dense-layer kernels: schedule selectors (the race screens compare the default schedule with a plain one) --------- */
/* dca_f16x3_gemm */
/* test hook: 3 (default) = 256 x 256 tiles filled by LDS-DMA on the ping-pong / half-tile schedule (two wave groups one barrier
 * apart, the DMA queue never drained); 2 = the same tile with two whole-K-step stages and one drain + barrier per K-step
 * (bit-identical to 3: same products in the same order — what the race screens in tests/ compare against).  ldo % 4 == 0,
 * out_h / out_l 8-byte and x_out / skip 16-byte aligned. */
int dca_f16x3_gemm_variant(int variant);

/* dca_gemm16 */
/* test hook: 3 (default) = the 8-phase schedule (two wave groups one barrier apart, half-tile staging, the DMA queue never
 * drained) with the MFMA operand roles swapped and a lean tail compiled per layer form — relu(a . w^T + bias (+ skip)) on whole
 * 256 x 256 tiles with 16-byte aligned rows; other forms and the ragged strips of a layer run on 2 = the same schedule with the
 * general tail; 1 = two whole K-step stages with one drain + barrier per K-step (the plain reference of the race screens).
 * Bit-identical results (same products, same accumulation order). */
int dca_gemm16_variant(int variant);

#ifdef __cplusplus
}
#endif
#endif /* DCA_DEBUG_H */
