// dca_engine.hip — device-resident batched weighted A* (BWAS) for gfx950 (MI355X).
//
// Replaces the reference's two search cores behind one C ABI (include/dca.h):
//   search_methods/astar.py:50-90,232-340      Instance / AStar.step   (DCA_SEM_PY)
//   cpp/parallel_weighted_astar.cpp:138-346    parallelWeightedAStar   (DCA_SEM_CPP)
//
// Everything the search touches lives in HBM:
//   node pool   SoA: state[N][D] u8 rows, g[N] i32, parent[N] u32, move[N] u8, solved[N] u8.
//               A node id is handed to EVERY generated child (id = base + pop_rank*A + move), so
//               the expansion kernel writes child rows straight to their final place, ids increase
//               in the reference's push order (astar.py:64-67 heappush_count) and (cost, id) is the
//               reference's (cost, count) FIFO tie-break.
//   CLOSED      open-addressing table of 16-byte slots {tag32|rep32, best g, batch list head},
//               keyed by the 64-bit state hash, verified against the representative node's state
//               bytes (exact key equality, like State.__eq__ / NodePointerEq).
//   OPEN        (cost key u64 = order-preserving bits of the f64 cost, node id u32 | is_solved flag) arrays in two tiers:
//               FRONT (entries with key <= T, edited IN PLACE: pops tombstone what they take, pushes append) and BACK
//               (the rest, append-only with tombstones).  pop = exact top-B of FRONT by (cost, id): an incrementally
//               maintained 4096-bin histogram gives the threshold bin — every workgroup of k_sel_collect scans it for
//               itself (k_sel_scan only runs in rebase iterations); every entry at or below the threshold bin is moved,
//               grouped by bin, into a scratch array; k_rank orders each bin exactly — bins of up to 512 entries a thread
//               per entry, larger ones bucketed in LDS on 64-bit key / id offsets and shared between workgroups — and
//               puts the overshoot of the threshold bin back into the slots it came from.  A threshold bin too large for
//               that (massive cost ties: an integer-valued heuristic makes every f-level one tie group of up to millions
//               of entries) is first cut down IN PLACE by k_sel_collect's whole grid: radix selection on the 96-bit
//               (key, id) composite, counter grid barriers between its passes (collect_giant).  Every 8th ("rebase")
//               iteration one pass compacts FRONT into its second buffer, evicts what a spill left above T, recounts the
//               histogram under a fresh binning; refills from BACK and spills to it are decided there, with hysteresis,
//               so an iteration costs O(|FRONT| + children), independent of |OPEN|.
//
// One BWAS iteration = pop -> expand -> heuristic -> dedup -> push in FOUR launches (collect, rank, expand with the CLOSED
// probe inside, commit; k_sel_scan's one-workgroup scan in front of them, four more in a rebase iteration), all stream-ordered, no host round trip: counts live in a device control block (hot
// counters on their own cache lines) and every kernel sizes itself from it; the last workgroup of k_commit closes the
// iteration and leaves the size of the next batch.  The per-iteration batch geometry is double buffered by iteration parity
// (IterState), so the expansion launch itself closes the pop (no single-thread kernel in between).  run_builtin replays
// chunks of up to 64 iterations as ONE hipGraph each (cached by their pattern of rebase iterations).  One engine steps K independent instances at once: every kernel
// takes the device array of instance descriptors and picks its instance with blockIdx.y.
//
// Sequential-order dedup, done in parallel: children of one batch that hit the same
// CLOSED slot are chained through the slot's `head`; child j is kept iff g_j < v0 (the slot's value
// before the batch, recorded by the probe) and no earlier chain member has g <= g_j — exactly astar.py:78-90 /
// cpp:244-265.  A child whose chain has a single member (the common case, flagged by the probe) decides from its
// own registers inside the push kernel.
#include <string.h>

#include <new>
#include <vector>

#include "dca_common.h"
#include "dca_tile.h"

namespace dca {

constexpr int NBIN = 4096;            // bins of the selection histogram (2048 left 13 000-entry threshold bins in long runs)
constexpr int kBinsPerThread = NBIN / 1024;  // of the single-workgroup scans
constexpr int kLgNbin = 12;
constexpr int kSub = 2048;            // sub-bins of k_rank's per-bin bucketing
constexpr int kScanBlocks = 256;      // grid of the OPEN scans: few fat blocks (cheap when they early-exit)
constexpr int kCollectBlocks = 512;   // k_sel_collect: two workgroups per CU keep twice the loads in flight
constexpr int kRankBlocks = 256;      // k_rank: one 512-thread workgroup per CU, the work units strided over them
constexpr int kTinyBinDefault = 512;  // bins up to this size are ranked one THREAD per entry (all-pairs inside the bin): a bin of its
                                      // own workgroup costs a chain of ~6 memory trips however small it is (g_tune[9] overrides)
constexpr int kSortCap = 8192;        // diagnostics only: bins beyond this many entries are counted as "giant"
constexpr int RT = 512;                            // threads of a k_rank workgroup (8 waves: up to 256 VGPRs each)
constexpr int kRegEnt = 16;                        // entries a thread keeps in registers
constexpr uint32_t kLdsEnt = RT * kRegEnt;         // items up to this size are bucketed entirely in LDS (96 KB)
constexpr int kStash = 2560;          // per-workgroup LDS stash of k_sel_collect (entries at or below the threshold bin)
// Segments k_rank may see: the selection bins; in "giant" iterations (the threshold bin holds more entries than k_rank can
// bucket in LDS — massive cost ties) k_sel_collect refines that bin across the whole grid first and hands over, in its
// place, the part that is certainly in the batch (segment bstar) and the last refinement level's sub-bins up to the one
// holding the batch's last entry (segments bstar+1 ...).
constexpr int kSegs = NBIN + 1 + kSub;
constexpr uint32_t kGiantBinDefault = 8192;   // = kLdsEnt: larger threshold bins are refined by the grid, not by one workgroup
constexpr int kMaxLevels = 9;                 // 96 composite bits / 11 bits per level
constexpr unsigned long long kBarrierTimeout = 200000000ull;  // 2 s of the 100 MHz wall clock: a stuck grid barrier fails the search
// The FIRST barrier of a giant iteration doubles as the residency check: nothing of the search has been modified before it, so
// a launch whose workgroups are not all resident (another process or stream holds CUs / LDS) gives the path up there — by
// consensus, see collect_grid_barrier — and takes the single-workgroup streaming path instead, for the rest of the engine's life.
constexpr unsigned long long kBarrierTimeoutFirst = 25000000ull;  // 0.25 s (a resident grid passes it in well under a millisecond)
constexpr uint32_t GBAR_BROKEN = 0x80000000u;  // bit 31 of the arrival counter: the barrier was abandoned before it completed
constexpr uint32_t NIL = 0xFFFFFFFFu;
// OPEN entries carry "this node is solved" in bit 31 of the id (node ids stay below 2^31): the pop then knows a goal
// without touching the node pool.  Every compare / index uses the id with the flag masked off.
constexpr uint32_t ID_MASK = 0x7FFFFFFFu, ID_SOLVED = 0x80000000u;
constexpr uint64_t EMPTY = ~0ull;
constexpr uint64_t DEAD = ~0ull;  // tombstone key of an OPEN entry that left its tier (BACK -> FRONT, FRONT -> batch / BACK)
constexpr uint32_t GINF = 0xFFFFFFFFu;
constexpr int kMaxMoves = 4096;

struct Slot {
    uint64_t entry;  // (hash >> 32) << 32 | representative node id ; EMPTY = unused
    uint32_t g;      // best path cost recorded for this state (GINF = none yet)
    uint32_t head;   // node id of the most recently chained child; < batch base = stale
};

// device control block: every count the kernels need, so nothing is read back by the host.
// Words that take atomics from many workgroups each sit on their own 128-byte line: same-address (and
// same-line) atomics retire at ~90 per microsecond on this chip, and they would otherwise also stall
// every wave that merely reads the per-iteration parameters next to them.
struct alignas(128) Cnt {
    uint32_t v;
    uint32_t pad[31];
};
struct alignas(128) GRange {
    uint64_t kmin, kmax;
    uint32_t imin, imax;
    uint32_t pad[26];
};
struct alignas(128) Rng {
    uint64_t kmin, kmax;
    uint64_t pad[14];
};
// What one iteration hands to the next, double buffered by iteration parity: kernels up to and including the
// expansion read S[iters & 1]; workgroup 0 of the expansion writes S[(iters + 1) & 1] (nobody reads that copy during
// the launch), the dedup / push kernels of the same iteration read it, and the push kernel's last workgroup makes it
// current by incrementing `iters`.
struct IterState {
    uint32_t pool_n;        // node ids handed out
    uint32_t npop, m, base; // batch of the iteration that produced this state
    uint32_t best_id;       // cpp: cheapest solved node popped so far
    int32_t has_best;
    float best_cost;
};
struct Ctl {
    // ---- read-mostly parameters, written by single-thread / single-workgroup code ----------------
    int32_t done, failed, stop_after, pad0;
    int64_t iters, gen, expanded;
    IterState S[2];
    // OPEN is two tiers of (key,id) arrays.  FRONT holds every entry with key <= T, BACK (buffer 2/3) the rest; pops
    // only ever look at FRONT, so an iteration costs O(|FRONT| + children), independent of |OPEN|.  FRONT is edited IN
    // PLACE: a pop tombstones the entries it takes (key = DEAD), pushes append; every kRefillPeriod-th iteration a
    // rebase pass squeezes the tombstones out into the other FRONT buffer (and recounts the selection histogram).
    uint32_t cur_f;         // the live FRONT buffer (0/1); the other one is the target of the next compaction
    uint32_t hbin;          // E.hist is kept up to date below this bin only (see k_sel_scan)
    uint32_t front_above;   // FRONT entries above T since the last spill: they leave for BACK in the next rebase pass
    uint32_t cur_b;         // the BACK buffer (2/3)
    uint64_t T;             // tier threshold key (inclusive upper bound of FRONT)
    uint32_t refill, compact, r_bstar, spill_bin, r_move;
    uint64_t r_kmin;
    uint32_t r_shift;
    // selection
    uint32_t want, bstar, shift, n_big, n_ord;
    uint32_t giant;   // this pop's threshold bin is refined across the grid by k_sel_collect (set by k_sel_scan)
    uint32_t tseg;    // the segment holding the batch's last entry (= bstar unless giant)
    uint32_t pre_b, cn_star;  // entries below the threshold bin / in it (k_rank's histogram writeback)
    uint64_t sel_kmin;
    // goals
    uint32_t goal_id;
    // diagnostics of the last pop (dca_engine_debug): entries handed to k_rank, largest bin among them, bins beyond
    // the LDS sort capacity (those take the refinement path)
    uint32_t dbg_nord, dbg_maxbin, dbg_giant, dbg_giant_seen, dbg_maxsub;
    // ---- hot words -----------------------------------------------------------------------------
    Cnt open_n[4];   // physical entries per OPEN buffer (0/1 FRONT + its compaction target, 2/3 BACK + its compaction target)
    Cnt closed_n, back_dead, front_dead, ret_n, ticket_a;  // *_dead: tombstones (key == DEAD) inside the tier's buffer
    Cnt gbar;        // arrivals at k_sel_collect's grid barrier (giant iterations; zeroed by k_sel_scan)
    GRange grange;   // key / id range of the giant threshold bin (atomics from every workgroup of k_sel_collect)
    Rng rng[4];      // running key range per OPEN buffer
    alignas(128) unsigned long long goal_best;  // PY: min over solved popped of (g << 32 | pop rank)
    alignas(128) uint32_t first_solved;         // CPP: smallest pop rank holding a solved node
};

__device__ __forceinline__ uint64_t key_of_cost(double c) {
    uint64_t b = (uint64_t)__double_as_longlong(c);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);  // order-preserving
}
__device__ __forceinline__ double cost_of_key(uint64_t k) {
    uint64_t b = (k & 0x8000000000000000ull) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)b);
}
__device__ __forceinline__ uint32_t select_shift(uint64_t kmin, uint64_t kmax) {
    uint64_t range = kmax - kmin;
    int bits = range ? 64 - __clzll((long long)range) : 0;
    return bits > kLgNbin ? (uint32_t)(bits - kLgNbin) : 0u;
}
// selection bin of a key under the binning (kmin, shift) in force: keys at or below kmin share bin 0, keys past the last
// bin share bin NBIN - 1 (the binning is only refreshed every kRefillPeriod iterations; children that land outside it
// meanwhile are still ordered exactly — k_rank works on each bin's own composite range)
__device__ __forceinline__ uint32_t bin_of(uint64_t k, uint64_t kmin, uint32_t shift) {
    if (k <= kmin) return 0u;
    const uint64_t f = (k - kmin) >> shift;
    return f < (uint64_t)NBIN ? (uint32_t)f : (uint32_t)(NBIN - 1);
}
__device__ __forceinline__ bool pair_less(uint64_t ka, uint32_t ia, uint64_t kb, uint32_t ib) {
    return ka < kb || (ka == kb && (ia & ID_MASK) < (ib & ID_MASK));
}

// launch slots of the device-side profile (dca_engine_profile_builtin)
enum {
    P_REFILL_HIST = 0, P_REFILL_SCAN, P_REFILL_MOVE, P_SEL_HIST, P_SEL_SCAN, P_SEL_COLLECT, P_RANK, P_EXPAND, P_PROBE,
    P_DECIDE, P_PACK, P_COMMIT, P_RANK_SMALL, P_RANK_BIG, P_RB_LOAD, P_RB_COUNT, P_RB_SCATTER, P_RB_ORDER, P_COUNT
};
constexpr int kProfSlots = 1024;
// diagnostics: knobs a tuning run can flip without a rebuild (dca_debug_tune); 0 = the shipped behaviour
__device__ int g_tune[16];
#define kTinyBin ((uint32_t)(g_tune[9] > 0 ? g_tune[9] : kTinyBinDefault))
// workgroups of k_rank that share one large bin (each reads and counts all of it, then scatters / orders / emits its own run of
// sub-bins): one per 1024 entries, at most 8 — the callers clamp.  (One per 512 or 256 entries — 3 to 6 workgroups on the
// 1250-1950-entry threshold bin instead of 2 — changes nothing: the bin's workgroups spend their time in the chain load ->
// count -> scatter -> order, not in the share of the scatter that shrinks; profiles/r05_engine_ab.txt.)
__device__ __forceinline__ uint32_t rank_shares(uint32_t cn) {
    return cn <= (uint32_t)(512 * 16) ? (cn + 1023u) / 1024u : 1u;  // (512 * 16 = kLdsEnt: larger bins stream through one workgroup)
}

struct Eng {
    int env, dim, D, A, B, sem, oh_dtype, depth;
    double w;
    float wf;
    uint32_t max_nodes, M, tab_cap, tab_mask;
    uint32_t front_cap;  // entries a FRONT buffer holds: max_nodes + room for the tombstones of kRefillPeriod iterations
    uint8_t* state;
    int32_t* g;
    uint32_t* parent;
    uint8_t* move;
    uint8_t* solved;
    Slot* tab;
    uint32_t* closed_slots;  // [max_nodes] the CLOSED slots in use, in insertion order: a reset clears these instead of the whole table
    uint64_t* open_key[4];
    uint32_t* open_id[4];
    uint32_t f_keep, f_max;  // FRONT hysteresis: refill/spill down to ~f_keep, spill when above f_max
    uint32_t *hist, *pre, *fill;  // selection histogram, its exclusive prefix [NBIN+1], per-bin fill of the scratch array
    uint32_t* rhist;              // histogram of BACK (refill), separate: `hist` is maintained across iterations
    uint32_t* subhist;            // [kMaxLevels][kSub] sub-bin counts of a giant threshold bin, one array per refinement level
    int coop;                     // k_sel_collect's grid is fully resident (single-instance engine): grid barriers allowed
    uint32_t* coop_off;           // device word, set once a grid barrier found the launch NOT fully resident (the GPU is shared): giant
                                  // bins go to k_rank's streaming path from then on.  Outside the control block: resets do not touch it
    uint64_t* part;  // [2][kCollectBlocks] per-block key ranges (min, max) of the entries k_front_rebase kept
    // scratch of the pop: every FRONT entry at or below the threshold bin, grouped by bin (bin f occupies
    // [pre[f], pre[f+1])), and — only for bins too large for LDS — the single-workgroup sub-bin ordering
    uint64_t* tmp_key;
    uint32_t* tmp_id;
    uint16_t* tmp_f;   // bin of every scratch entry (k_rank's thread-per-entry pass over the small bins)
    uint32_t* tmp_idx;  // FRONT position each scratch entry was taken from (k_rank hands entries back into those slots)
    uint64_t* ord_key;
    uint32_t* ord_id;
    uint32_t* big_list;  // work units of k_rank's large-bin pass: the segments at or below the threshold with more than kTinyBin
                         // entries, one unit per workgroup that shares the segment (segment | share << 16 | shares << 20)
    uint64_t* pop_key;   // the batch in pop order
    uint32_t *pop_id, *pop_g;
    uint64_t* child_hash;
    uint32_t *child_slot, *child_next, *child_v0;
    uint8_t *child_flags, *child_multi;
    float* child_h;
    uint8_t* nnet_in;
    uint8_t* onehot;
    uint8_t* root_nnet;
    int32_t* d_moves;
    Ctl* ctl;
    unsigned long long* prof;  // [P_COUNT][kProfSlots][2] device wall-clock stamps (min start, max end) or null
    // dedup-first ("packed") stepping: only the children that survive the CLOSED check reach the heuristic
    uint32_t* kept_pos;   // [M] row of child j in the packed batch (NIL = dropped)
    uint32_t* pk_n;       // rows packed this iteration, shared by every instance of the engine
    uint32_t* pk_src;     // [K*M] packed row -> instance*M + child index
    uint8_t* pk_nnet;     // [K*M, D] network-input rows of the kept children
    uint8_t* pk_onehot;   // [K*M, pk_stride] one-hot rows (pk_dtype), tail of each row zero
    const float* pk_h;    // [K*M] heuristic of the packed rows (written by the caller between the two halves)
    uint32_t pk_stride, inst;
    int pk_dtype;
};

__device__ __forceinline__ const IterState& st_cur(const Ctl* c) { return c->S[c->iters & 1]; }
__device__ __forceinline__ const IterState& st_next(const Ctl* c) { return c->S[(c->iters + 1) & 1]; }

// the binning a rebase iteration installs (k_front_rebase recounts under it, k_sel_scan records it): FRONT's exact key range,
// its top raised to the tier threshold — no key above T enters FRONT before the next rebase
__device__ __forceinline__ void fresh_binning(const Ctl* c, uint32_t buf, uint64_t& kmin, uint32_t& shift) {
    kmin = c->rng[buf].kmin;
    uint64_t kmax = c->rng[buf].kmax;
    const uint64_t T = c->T;
    if (T != ~0ull && T > kmax) kmax = T;
    shift = kmax > kmin ? select_shift(kmin, kmax) : 0u;
}

// Device-side profile: thread 0 of every workgroup (instance 0 only) folds its wall-clock start / end into the
// launch's slot, so the host can read each launch's busy span and the gap to the next one INSIDE a replayed hipGraph.
// Off (prof == nullptr) it costs one scalar compare.
// (every instance's workgroups stamp into the same slots: with K instances sharing a launch the span is the launch's envelope)
struct Stamp {
    unsigned long long* p;
    unsigned long long t0;
    __device__ __forceinline__ Stamp(const Eng& E, int kid) : p(nullptr), t0(0) {
        if (E.prof != nullptr && threadIdx.x == 0) {
            p = E.prof + ((size_t)kid * kProfSlots + (blockIdx.x & (kProfSlots - 1))) * 2;
            t0 = wall_clock64();
        }
    }
    __device__ __forceinline__ ~Stamp() {
        if (p) {
            atomicMin(p, t0);
            atomicMax(p + 1, (unsigned long long)wall_clock64());
        }
    }
};

// phase marks inside a launch (same slots; profile only)
__device__ __forceinline__ void prof_begin(const Eng& E, int kid) {
    if (E.prof != nullptr && threadIdx.x == 0)
        atomicMin(E.prof + ((size_t)kid * kProfSlots + (blockIdx.x & (kProfSlots - 1))) * 2, (unsigned long long)wall_clock64());
}
__device__ __forceinline__ void prof_end(const Eng& E, int kid) {
    if (E.prof != nullptr && threadIdx.x == 0)
        atomicMax(E.prof + ((size_t)kid * kProfSlots + (blockIdx.x & (kProfSlots - 1))) * 2 + 1, (unsigned long long)wall_clock64());
}

// ---------------------------------------------------------------------------------------------
// wave-aggregated append to the live OPEN buffer (+ running min/max of its keys)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void open_append(const Eng& E, Ctl* c, uint32_t buf, bool pred, uint64_t key, uint32_t id) {
    unsigned long long mask = __ballot(pred);
    if (mask == 0) return;
    int lane = threadIdx.x & 63;
    int leader = __ffsll((long long)mask) - 1;
    uint32_t cnt = (uint32_t)__popcll(mask);
    uint32_t basep = 0;
    // wave min / max of the appended keys
    uint64_t kmn = pred ? key : ~0ull, kmx = pred ? key : 0ull;
    for (int o = 32; o > 0; o >>= 1) {
        uint64_t a = __shfl_xor(kmn, o), b = __shfl_xor(kmx, o);
        kmn = a < kmn ? a : kmn;
        kmx = b > kmx ? b : kmx;
    }
    if (lane == leader) {
        basep = atomicAdd(&c->open_n[buf].v, cnt);
        atomicMin((unsigned long long*)&c->rng[buf].kmin, (unsigned long long)kmn);
        atomicMax((unsigned long long*)&c->rng[buf].kmax, (unsigned long long)kmx);
    }
    basep = __shfl(basep, leader);
    if (pred) {
        uint32_t pos = basep + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (pos < (buf < 2 ? E.front_cap : E.max_nodes)) {
            E.open_key[buf][pos] = key;
            E.open_id[buf][pos] = id;
        } else {
            c->failed = 1;
        }
    }
}

// block-aggregated reservation in two output arrays at once: every thread asks for cntA / cntB slots,
// the block issues ONE atomicAdd per array (same-address atomics saturate near 90 per microsecond on
// this chip, so per-wave appends would bound a multi-million-entry pass).  sh: 2*NW+2 words of LDS.
struct Pos2 {
    uint32_t a, b;
};
template <int NT>
__device__ __forceinline__ Pos2 block_reserve2(uint32_t cntA, uint32_t cntB, uint32_t* ctrA, uint32_t* ctrB,
                                               uint32_t* sh) {
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t ia = cntA, ib = cntB;
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t va = __shfl_up(ia, o), vb = __shfl_up(ib, o);
        if (lane >= o) {
            ia += va;
            ib += vb;
        }
    }
    if (lane == 63) {
        sh[wv] = ia;
        sh[NW + wv] = ib;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t aa = 0, ab = 0;
        for (int w = 0; w < NW; w++) {
            uint32_t ta = sh[w], tb = sh[NW + w];
            sh[w] = aa;
            sh[NW + w] = ab;
            aa += ta;
            ab += tb;
        }
        sh[2 * NW] = aa ? atomicAdd(ctrA, aa) : 0u;
        sh[2 * NW + 1] = ab ? atomicAdd(ctrB, ab) : 0u;
    }
    __syncthreads();
    Pos2 p{sh[2 * NW] + sh[wv] + ia - cntA, sh[2 * NW + 1] + sh[NW + wv] + ib - cntB};
    __syncthreads();
    return p;
}

struct Pos3 {
    uint32_t a, b, c;
};
template <int NT>
__device__ __forceinline__ Pos3 block_reserve3(uint32_t cntA, uint32_t cntB, uint32_t cntC, uint32_t* ctrA,
                                               uint32_t* ctrB, uint32_t* ctrC, uint32_t* sh /*3*NW+3*/) {
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t ia = cntA, ib = cntB, ic = cntC;
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t va = __shfl_up(ia, o), vb = __shfl_up(ib, o), vc = __shfl_up(ic, o);
        if (lane >= o) {
            ia += va;
            ib += vb;
            ic += vc;
        }
    }
    if (lane == 63) {
        sh[wv] = ia;
        sh[NW + wv] = ib;
        sh[2 * NW + wv] = ic;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t aa = 0, ab = 0, ac = 0;
        for (int w = 0; w < NW; w++) {
            uint32_t ta = sh[w], tb = sh[NW + w], tc = sh[2 * NW + w];
            sh[w] = aa;
            sh[NW + w] = ab;
            sh[2 * NW + w] = ac;
            aa += ta;
            ab += tb;
            ac += tc;
        }
        sh[3 * NW] = aa ? atomicAdd(ctrA, aa) : 0u;
        sh[3 * NW + 1] = ab ? atomicAdd(ctrB, ab) : 0u;
        sh[3 * NW + 2] = ac ? atomicAdd(ctrC, ac) : 0u;
    }
    __syncthreads();
    Pos3 p{sh[3 * NW] + sh[wv] + ia - cntA, sh[3 * NW + 1] + sh[NW + wv] + ib - cntB,
           sh[3 * NW + 2] + sh[2 * NW + wv] + ic - cntC};
    __syncthreads();
    return p;
}

// K-way variant: cnt[k] slots wanted in the array behind ctr[k]; pos[k] receives the first slot
template <int NT, int K>
__device__ __forceinline__ void block_reserveK(const uint32_t (&cnt)[K], uint32_t* const (&ctr)[K], uint32_t (&pos)[K],
                                               uint32_t* sh /*K*NW+K*/) {
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc[K];
#pragma unroll
    for (int k = 0; k < K; k++) inc[k] = cnt[k];
    for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            uint32_t v = __shfl_up(inc[k], o);
            if (lane >= o) inc[k] += v;
        }
    }
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < K; k++) sh[k * NW + wv] = inc[k];
    }
    __syncthreads();
    if (threadIdx.x < K) {
        const int k = threadIdx.x;
        uint32_t acc = 0;
        for (int w = 0; w < NW; w++) {
            uint32_t tv = sh[k * NW + w];
            sh[k * NW + w] = acc;
            acc += tv;
        }
        sh[K * NW + k] = acc ? atomicAdd(ctr[k], acc) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) pos[k] = sh[K * NW + k] + sh[k * NW + wv] + inc[k] - cnt[k];
    __syncthreads();
}

// fold a thread's running key range into a buffer's kmin/kmax (one atomic per wave, only if it helps)
__device__ __forceinline__ void fold_range(Ctl* c, uint32_t buf, uint64_t kmn, uint64_t kmx) {
    for (int o = 32; o > 0; o >>= 1) {
        uint64_t a = __shfl_xor(kmn, o), b = __shfl_xor(kmx, o);
        kmn = a < kmn ? a : kmn;
        kmx = b > kmx ? b : kmx;
    }
    if ((threadIdx.x & 63) == 0) {
        if (kmn < c->rng[buf].kmin) atomicMin((unsigned long long*)&c->rng[buf].kmin, (unsigned long long)kmn);
        if (kmx > c->rng[buf].kmax) atomicMax((unsigned long long*)&c->rng[buf].kmax, (unsigned long long)kmx);
    }
}

// ---------------------------------------------------------------------------------------------
// reset / root
// ---------------------------------------------------------------------------------------------
// CLOSED is cleared before a search starts.  The table is sized for max_nodes (`--max_nodes auto`: ~1e9 ids, a 32 GiB table),
// most searches touch a sliver of it: k_commit records the slot of every state it inserts (E.closed_slots, in insertion
// order), and a reset clears THOSE — unless the last search used more than 1/8 of the table (a streaming clear is cheaper than
// random 16-byte stores then), the table has never been cleared, or the last iteration was abandoned between its probe and
// its commit (`force`: slots were claimed that the list does not hold yet).  Both kernels are enqueued; each one looks at the
// previous search's count (still in the control block: k_reset runs behind them) and one of them returns at once.
__device__ __forceinline__ bool clear_by_list(const Ctl* c, uint32_t cap, int force) {
    return !force && (uint64_t)c->closed_n.v * 8ull < (uint64_t)cap;
}
__global__ void k_init_table(Slot* tab, uint32_t cap, const Ctl* c, int force) {
    if (clear_by_list(c, cap, force)) return;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t stride = gridDim.x * blockDim.x;
    for (; i < cap; i += stride) {
        tab[i].entry = EMPTY;
        tab[i].g = GINF;
        tab[i].head = 0;
    }
}
__global__ void k_clear_table_list(Slot* tab, uint32_t cap, const uint32_t* __restrict__ slots, const Ctl* c, int force) {
    if (!clear_by_list(c, cap, force)) return;
    const uint32_t n = c->closed_n.v;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t sl = slots[i];
        if (sl < cap) {
            tab[sl].entry = EMPTY;
            tab[sl].g = GINF;
            tab[sl].head = 0;
        }
    }
}

__global__ void k_reset(Eng E) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Ctl* c = E.ctl;
    memset(c, 0, sizeof(Ctl));
    const uint8_t* s = E.state;  // node 0 = root, already copied in
    uint64_t h = hash_init(E.D);
    bool ok = true;
    for (int k = 0; k < E.D; k += 8) {
        uint64_t w = 0;
        for (int j = 0; j < 8 && k + j < E.D; j++) {
            uint32_t b = s[k + j];
            uint32_t goal = goal_byte(E.env, E.D, k + j);
            ok &= (b == goal);
            w |= (uint64_t)b << (8 * j);
            E.root_nnet[k + j] = (uint8_t)(E.env == DCA_ENV_CUBE3 ? (b * 57u) >> 9 : b);
        }
        h = hash_word(h, w);
    }
    h = hash_final(h);
    E.g[0] = 0;
    E.parent[0] = NIL;
    E.move[0] = 0xFF;
    E.solved[0] = ok ? 1 : 0;
    for (int p = 0; p < 2; p++) c->S[p].pool_n = 1;
    c->cur_f = 0;
    c->goal_best = ~0ull;
    c->first_solved = NIL;
    for (int b = 0; b < 4; b++) {
        c->rng[b].kmin = ~0ull;
        c->rng[b].kmax = 0;
    }
    c->cur_b = 2;
    c->T = ~0ull;  // everything is FRONT until the first spill
    if (E.sem == DCA_SEM_CPP) {
        // cpp:160-166: root pushed with cost 0 (h never evaluated), a copy inserted in CLOSED, gen = 1
        uint32_t slot = (uint32_t)h & E.tab_mask;
        E.tab[slot].entry = ((h >> 32) << 32) | 0u;
        E.tab[slot].g = 0;
        E.closed_slots[0] = slot;
        c->closed_n.v = 1;
        c->gen = 1;
        uint64_t key = key_of_cost(0.0);
        E.open_key[0][0] = key;
        E.open_id[0][0] = ok ? ID_SOLVED : 0u;
        c->open_n[0].v = 1;
        c->rng[0].kmin = c->rng[0].kmax = key;
    }
}

__global__ void k_root_commit(Eng E, const float* h_root) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Ctl* c = E.ctl;
    if (E.sem != DCA_SEM_PY || c->open_n[c->cur_f].v != 0) return;
    // astar.py:246-249,196: cost = w*0.0 + max(h,0)*!solved   (float64)
    double hv = fmax((double)h_root[0], 0.0);
    double cost = __dadd_rn(__dmul_rn(E.w, 0.0), __dmul_rn(hv, E.solved[0] ? 0.0 : 1.0));
    uint64_t key = key_of_cost(cost);
    uint32_t b = c->cur_f;
    E.open_key[b][0] = key;
    E.open_id[b][0] = E.solved[0] ? ID_SOLVED : 0u;
    c->open_n[b].v = 1;
    c->rng[b].kmin = c->rng[b].kmax = key;
}

// ---------------------------------------------------------------------------------------------
// refill: when FRONT holds fewer than one batch, raise T and move the cheapest part of BACK over
// (rare: amortised over the iterations FRONT then lasts).  Always enqueued; exits early when idle.
// ---------------------------------------------------------------------------------------------
// The refill kernels are only enqueued every kRefillPeriod-th iteration (they cost a few microseconds
// even when idle), so they top FRONT up early enough to last until the next check: an iteration
// removes at most B entries from FRONT.
constexpr int kRefillPeriod = 16;
// what a refill / spill must leave in FRONT so that it cannot run short before the next refill check
__device__ __forceinline__ uint32_t front_keep(const Eng& E) {
    const uint32_t floor_ = (uint32_t)(kRefillPeriod + 1) * (uint32_t)E.B;
    return E.f_keep > floor_ ? E.f_keep : floor_;
}
__device__ __forceinline__ bool need_refill(const Eng& E, const Ctl* c) {
    return c->open_n[c->cur_b].v != c->back_dead.v &&
           c->open_n[c->cur_f].v - c->front_dead.v - c->front_above < (uint32_t)(kRefillPeriod + 1) * (uint32_t)E.B;
}

__global__ __launch_bounds__(256) void k_refill_hist(const Eng* __restrict__ engs) {
    const Eng& E = engs[blockIdx.y];
    Ctl* c = E.ctl;
    if (c->done || !need_refill(E, c)) return;
    Stamp stamp(E, P_REFILL_HIST);
    __shared__ uint32_t lh[NBIN];
    for (int i = threadIdx.x; i < NBIN; i += 256) lh[i] = 0;
    __syncthreads();
    const uint32_t b = c->cur_b, n = c->open_n[b].v;
    const uint64_t kmin = c->rng[b].kmin;
    const uint32_t shift = select_shift(kmin, c->rng[b].kmax);
    const uint64_t* __restrict__ keys = E.open_key[b];
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        uint64_t k = keys[i];
        if (k == DEAD) continue;
        uint64_t f = (k - kmin) >> shift;
        atomicAdd(&lh[f < NBIN ? (uint32_t)f : NBIN - 1], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NBIN; i += 256)
        if (lh[i]) atomicAdd(&E.rhist[i], lh[i]);
}

// exclusive prefix of NBIN bin counts held kBinsPerThread per thread (1024 threads) into pre[0..NBIN]
__device__ __forceinline__ void scan_vals(const uint32_t (&v)[kBinsPerThread], uint32_t* pre, uint32_t* wsum) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < kBinsPerThread; k++) s += v[k];
    uint32_t incl = s;
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t u = __shfl_up(incl, o);
        if (lane >= o) incl += u;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    if (t < 16) {
        uint32_t u = wsum[t], acc = u;
        for (int o = 1; o < 16; o <<= 1) {
            uint32_t x = __shfl_up(acc, o, 16);
            if (t >= o) acc += x;
        }
        wsum[t] = acc - u;  // exclusive wave offsets
    }
    __syncthreads();
    uint32_t run = incl - s + wsum[wv];
#pragma unroll
    for (int k = 0; k < kBinsPerThread; k++) {
        pre[kBinsPerThread * t + k] = run;
        run += v[k];
    }
    if (t == 1023) pre[NBIN] = run;
    __syncthreads();
}

// exclusive prefix of NBIN global bins into pre[0..NBIN] (1024 threads, kBinsPerThread bins each); optionally zeroes them
__device__ __forceinline__ void scan_bins(uint32_t* __restrict__ hist, bool zero, uint32_t* pre, uint32_t* wsum) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    uint32_t v[kBinsPerThread], s = 0;
#pragma unroll
    for (int k = 0; k < kBinsPerThread; k++) {
        v[k] = hist[kBinsPerThread * t + k];
        if (zero) hist[kBinsPerThread * t + k] = 0;
        s += v[k];
    }
    uint32_t incl = s;
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t u = __shfl_up(incl, o);
        if (lane >= o) incl += u;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    if (t < 16) {
        uint32_t u = wsum[t], acc = u;
        for (int o = 1; o < 16; o <<= 1) {
            uint32_t x = __shfl_up(acc, o, 16);
            if (t >= o) acc += x;
        }
        wsum[t] = acc - u;  // exclusive wave offsets
    }
    __syncthreads();
    uint32_t run = incl - s + wsum[wv];
#pragma unroll
    for (int k = 0; k < kBinsPerThread; k++) {
        pre[kBinsPerThread * t + k] = run;
        run += v[k];
    }
    if (t == 1023) pre[NBIN] = run;
    __syncthreads();
}

__global__ __launch_bounds__(1024) void k_refill_scan(const Eng* __restrict__ engs) {
    const Eng& E = engs[blockIdx.y];
    Ctl* c = E.ctl;
    if (c->done) return;
    Stamp stamp(E, P_REFILL_SCAN);
    // this launch opens a "rebase" iteration: k_front_rebase compacts and recounts FRONT from scratch under a fresh binning right after
    // the refill, so the incrementally maintained histogram is dropped here
    for (int k = 0; k < kBinsPerThread; k++) E.hist[kBinsPerThread * threadIdx.x + k] = 0;
    if (threadIdx.x == 0) c->open_n[c->cur_f ^ 1].v = 0;  // the rebase pass compacts FRONT into the other buffer
    if (!need_refill(E, c)) {
        if (threadIdx.x == 0) {
            // Nothing to move over — but spills (k_front_rebase) and pushes (k_commit) keep appending to BACK while its
            // tombstones stay: when the buffer nears its physical end, squeeze them out now (a compaction-only pass of
            // k_refill_move) instead of failing a search whose OPEN would fit.
            const uint32_t b = c->cur_b, phys = c->open_n[b].v, dead = c->back_dead.v;
            // (g_tune[2]: test hook — squeeze once BACK holds more than max_nodes * v / 1024 physical entries)
            const uint32_t mark = g_tune[2] > 0 ? (uint32_t)(((uint64_t)E.max_nodes * (uint32_t)g_tune[2]) >> 10) : (E.max_nodes / 8) * 7;
            const bool squeeze = dead != 0 && phys > mark;
            c->refill = squeeze ? 1u : 0u;
            c->compact = squeeze ? 1u : 0u;
            c->r_move = 0;
            if (squeeze) {
                c->open_n[b ^ 1].v = 0;
                c->rng[b ^ 1].kmin = c->rng[b].kmin;
                c->rng[b ^ 1].kmax = c->rng[b].kmax;
            }
        }
        return;
    }
    __shared__ uint32_t pre[NBIN + 1];
    __shared__ uint32_t wsum[16];
    scan_bins(E.rhist, true, pre, wsum);
    const int t = threadIdx.x;
    const uint32_t b = c->cur_b, n = c->open_n[b].v - c->back_dead.v;
    const uint32_t target = n < front_keep(E) ? n : front_keep(E);
    for (int k = 0; k < kBinsPerThread; k++) {
        int bin = kBinsPerThread * t + k;
        if (pre[bin] < target && target <= pre[bin + 1]) c->r_bstar = (uint32_t)bin;  // whole bins move
    }
    __syncthreads();
    if (t == 0) {
        const uint64_t kmin = c->rng[b].kmin;
        const uint32_t shift = select_shift(kmin, c->rng[b].kmax);
        c->refill = 1;
        c->r_move = 1;
        c->r_kmin = kmin;
        c->r_shift = shift;
        // new tier threshold = top key of the last bin that moves (the final bin also absorbs overflow)
        uint64_t top = (c->r_bstar >= NBIN - 1) ? ~0ull : kmin + (((uint64_t)c->r_bstar + 1) << shift) - 1;
        if (top < kmin) top = ~0ull;  // wrapped
        c->T = top;
        if (top != ~0ull) c->rng[b].kmin = top + 1;  // everything at or below `top` leaves BACK
        // BACK is append-only with tombstones; squeeze them out once they outnumber the live entries
        // (entries can re-enter BACK through spills, so its physical size is not bounded by the pool)
        const uint32_t phys = c->open_n[b].v, dead = c->back_dead.v;
        c->compact = ((dead > phys / 2 && phys > 4096) || (dead != 0 && phys > (E.max_nodes / 8) * 7)) ? 1u : 0u;
        if (c->compact) {
            c->open_n[b ^ 1].v = 0;
            c->rng[b ^ 1].kmin = c->rng[b].kmin;
            c->rng[b ^ 1].kmax = c->rng[b].kmax;
        }
    }
}

// moved entries are appended to FRONT and tombstoned in place: BACK is never compacted (its dead
// entries are only the ones that moved, a small fraction of what keeps arriving)
__global__ __launch_bounds__(256) void k_refill_move(const Eng* __restrict__ engs) {
    const Eng& E = engs[blockIdx.y];
    Ctl* c = E.ctl;
    if (c->done || !c->refill) return;
    Stamp stamp(E, P_REFILL_MOVE);
    __shared__ uint32_t sh[2 * 4 + 2];
    const uint32_t sb = c->cur_b, fb = c->cur_f, db = sb ^ 1;  // BACK buffers are 2 and 3
    const bool compact = c->compact != 0;  // also squeeze the tombstones out into the other BACK buffer
    const uint32_t n = c->open_n[sb].v;
    const uint64_t kmin = c->r_kmin;
    const uint32_t shift = c->r_shift, bstar = c->r_bstar;
    const bool move = c->r_move != 0;  // (false: compaction-only pass, nothing crosses over to FRONT)
    uint64_t* __restrict__ keys = E.open_key[sb];
    const uint32_t* __restrict__ ids = E.open_id[sb];
    constexpr uint32_t ITEMS = 8, TILE = 256 * ITEMS;
    uint64_t fmn = ~0ull, fmx = 0;
    uint32_t moved = 0;
    const uint32_t ntiles = (n + TILE - 1) / TILE;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint64_t k[ITEMS];
        uint32_t tof = 0, stay = 0, cf = 0, cs = 0;
#pragma unroll
        for (uint32_t i = 0; i < ITEMS; i++) {
            uint32_t idx = tile * TILE + i * 256 + threadIdx.x;
            k[i] = keys[idx < n ? idx : n - 1];  // unconditional (index-clamped) load: all eight stay in flight
            if (idx >= n) k[i] = DEAD;
            uint64_t f = (k[i] - kmin) >> shift;
            bool alive = k[i] != DEAD;
            bool front = alive && move && (f < NBIN ? (uint32_t)f : NBIN - 1) <= bstar;
            tof |= (front ? 1u : 0u) << i;
            stay |= ((alive && !front && compact) ? 1u : 0u) << i;
            cf += front ? 1u : 0u;
            cs += (alive && !front && compact) ? 1u : 0u;
        }
        Pos2 p = block_reserve2<256>(cf, cs, &c->open_n[fb].v, &c->open_n[db].v, sh);
#pragma unroll
        for (uint32_t i = 0; i < ITEMS; i++) {
            uint32_t idx = tile * TILE + i * 256 + threadIdx.x;
            if ((tof >> i) & 1u) {
                if (p.a < E.front_cap) {
                    E.open_key[fb][p.a] = k[i];
                    E.open_id[fb][p.a] = ids[idx];
                }
                if (!compact) keys[idx] = DEAD;
                p.a++;
                moved++;
                fmn = k[i] < fmn ? k[i] : fmn;
                fmx = k[i] > fmx ? k[i] : fmx;
            } else if ((stay >> i) & 1u) {
                E.open_key[db][p.b] = k[i];
                E.open_id[db][p.b] = ids[idx];
                p.b++;
            }
        }
    }
    fold_range(c, fb, fmn, fmx);
    if (!compact) {
        for (int o = 32; o > 0; o >>= 1) moved += __shfl_xor(moved, o);
        if ((threadIdx.x & 63) == 0 && moved) atomicAdd(&c->back_dead.v, moved);
    }
}

// ---------------------------------------------------------------------------------------------
// pop: exact top-B of FRONT by (cost key, node id)
// ---------------------------------------------------------------------------------------------
// S1: histogram of (key - kmin) >> shift over FRONT, 2048 bins, LDS-privatised
__global__ __launch_bounds__(256) void k_front_rebase(const Eng* __restrict__ engs) {
    // every kRefillPeriod-th iteration: squeeze FRONT's tombstones out (live entries -> the other FRONT buffer, one
    // reservation per 2048-entry tile), recount the selection histogram under a fresh binning and take the exact key
    // range of what is left (the next rebase bins by it)
    const Eng& E = engs[blockIdx.y];
    Ctl* c = E.ctl;
    if (c->done) return;
    Stamp stamp(E, P_SEL_HIST);
    __shared__ uint32_t lh[NBIN];
    __shared__ uint32_t sh[2 * 4 + 2];
    for (int i = threadIdx.x; i < NBIN; i += 256) lh[i] = 0;
    __syncthreads();
    const uint32_t b = c->cur_f, nb = b ^ 1, n = c->open_n[b].v;
    // entries above the tier threshold leave for BACK on the way (a spill lowers T and leaves the move to this pass; a
    // BACK compaction in flight — k_refill_move — continues in the other BACK buffer, which k_sel_scan makes current)
    const uint32_t bb = (c->refill && c->compact) ? c->cur_b ^ 1u : c->cur_b;
    const uint64_t T = c->T;
    uint64_t kmin;
    uint32_t shift;
    fresh_binning(c, b, kmin, shift);
    const uint64_t* __restrict__ keys = E.open_key[b];
    const uint32_t* __restrict__ ids = E.open_id[b];
    constexpr uint32_t ITEMS = 8, TILE = 256 * ITEMS;
    uint64_t fmn = ~0ull, fmx = 0, bmn = ~0ull, bmx = 0;
    const uint32_t ntiles = (n + TILE - 1) / TILE;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint64_t k[ITEMS];
        uint32_t id[ITEMS];
        uint32_t cl = 0, ce = 0;
#pragma unroll
        for (uint32_t i = 0; i < ITEMS; i++) {
            const uint32_t idx = tile * TILE + i * 256 + threadIdx.x, ic = idx < n ? idx : n - 1;
            k[i] = keys[ic];
            id[i] = ids[ic];
            if (idx >= n) k[i] = DEAD;
            cl += (k[i] != DEAD && k[i] <= T) ? 1u : 0u;
            ce += (k[i] != DEAD && k[i] > T) ? 1u : 0u;
        }
        Pos2 p = block_reserve2<256>(cl, ce, &c->open_n[nb].v, &c->open_n[bb].v, sh);
#pragma unroll
        for (uint32_t i = 0; i < ITEMS; i++) {
            if (k[i] != DEAD && k[i] > T) {
                if (p.b < E.max_nodes) {
                    E.open_key[bb][p.b] = k[i];
                    E.open_id[bb][p.b] = id[i];
                } else {
                    c->failed = 1;
                }
                p.b++;
                bmn = k[i] < bmn ? k[i] : bmn;
                bmx = k[i] > bmx ? k[i] : bmx;
                continue;
            }
            const bool lv = k[i] != DEAD;
            // (grouping equal bins of a wave by ballot before the LDS atomic measured slower: 33 vs 29 us)
            if (lv) atomicAdd(&lh[bin_of(k[i], kmin, shift)], 1u);
            if (!lv) continue;
            if (p.a < E.front_cap) {
                E.open_key[nb][p.a] = k[i];
                E.open_id[nb][p.a] = id[i];
            } else {
                c->failed = 1;
            }
            p.a++;
            fmn = k[i] < fmn ? k[i] : fmn;
            fmx = k[i] > fmx ? k[i] : fmx;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NBIN; i += 256)
        if (lh[i]) atomicAdd(&E.hist[i], lh[i]);
    if (T != ~0ull) fold_range(c, bb, bmn, bmx);
    {
        __shared__ uint64_t red[2][4];
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        for (int o = 32; o > 0; o >>= 1) {
            const uint64_t a = __shfl_xor(fmn, o), z = __shfl_xor(fmx, o);
            fmn = a < fmn ? a : fmn;
            fmx = z > fmx ? z : fmx;
        }
        if (lane == 0) {
            red[0][wv] = fmn;
            red[1][wv] = fmx;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t mn = red[0][0], mx = red[1][0];
            for (int w = 1; w < 4; w++) {
                mn = red[0][w] < mn ? red[0][w] : mn;
                mx = red[1][w] > mx ? red[1][w] : mx;
            }
            E.part[blockIdx.x] = mn;
            E.part[kCollectBlocks + blockIdx.x] = mx;
        }
    }
}

// S2: one workgroup — prefix over the bins, threshold bin, spill decision, per-iteration counter reset
__global__ __launch_bounds__(1024) void k_sel_scan(const Eng* __restrict__ engs, int rebased) {
    const Eng& E = engs[blockIdx.y];
    Ctl* c = E.ctl;
    __shared__ uint32_t pre[NBIN + 1];
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t s_spill, s_bstar, s_maxbin, s_giant, s_nbig, s_sp, s_hbin;
    __shared__ uint64_t s_red[2][16];
    const int t = threadIdx.x;
    // Everything this launch reads from the control block, the histogram and (rebase) the per-block key ranges is
    // requested here, in one batch: a read that waits for an earlier one costs a trip to memory each (the launch is a
    // single workgroup: nothing else hides it), and there used to be six such trips in a row.
    const uint32_t done = c->done;
    const uint32_t cur0 = c->cur_f, old_shift = c->shift, refill = c->refill, compact = c->compact;
    const uint64_t old_kmin = c->sel_kmin, T0 = c->T;
    const uint32_t on0 = c->open_n[0].v, on1 = c->open_n[1].v, dead0 = c->front_dead.v, above0 = c->front_above;
    const uint64_t r0min = c->rng[0].kmin, r0max = c->rng[0].kmax, r1min = c->rng[1].kmin, r1max = c->rng[1].kmax;
    const uint32_t giant_seen = c->dbg_giant_seen;
    uint32_t hv[kBinsPerThread];
#pragma unroll
    for (int k = 0; k < kBinsPerThread; k++) hv[k] = E.hist[kBinsPerThread * t + k];
    uint64_t mn = ~0ull, mx = 0;
    if (rebased && t < kCollectBlocks) {
        mn = E.part[t];
        mx = E.part[kCollectBlocks + t];
    }
    if (done) return;
    Stamp stamp(E, P_SEL_SCAN);
    uint32_t cb = cur0;
    uint64_t new_kmin = old_kmin;
    uint32_t new_shift = old_shift;
    if (rebased) {
        // k_front_rebase compacted FRONT into the other buffer, counted E.hist under this binning (fresh_binning of the
        // old buffer's key range) and left per-block key ranges of the entries it kept in E.part
        new_kmin = cur0 ? r1min : r0min;
        uint64_t kmax = cur0 ? r1max : r0max;
        if (T0 != ~0ull && T0 > kmax) kmax = T0;
        new_shift = kmax > new_kmin ? select_shift(new_kmin, kmax) : 0u;
        for (int o = 32; o > 0; o >>= 1) {
            const uint64_t a = __shfl_xor(mn, o), z = __shfl_xor(mx, o);
            mn = a < mn ? a : mn;
            mx = z > mx ? z : mx;
        }
        if ((t & 63) == 0) {
            s_red[0][t >> 6] = mn;
            s_red[1][t >> 6] = mx;
        }
        cb ^= 1;
    }
    if (t == 0) {
        s_maxbin = 0;
        s_giant = 0;
        s_nbig = 0;
        s_spill = NBIN;  // no spill
        s_bstar = 0;
        s_hbin = NBIN;
    }
    // E.hist is FRONT's histogram under the binning in force: recounted by k_front_rebase in a rebase iteration (every
    // kRefillPeriod-th), maintained incrementally in between (the writeback below + k_commit's pushes)
    scan_vals(hv, pre, wsum);
    // live entries at or below T (a rebase pass has just dropped the tombstones and moved what was above T to BACK)
    const uint32_t n = rebased ? (cb ? on1 : on0) : (cb ? on1 : on0) - dead0 - above0;
    const uint32_t want = n < (uint32_t)E.B ? n : (uint32_t)E.B;
    const uint32_t hwant = (uint32_t)(kRefillPeriod + 2) * (uint32_t)E.B;
    for (int k = 0; k < kBinsPerThread; k++) {
        int bin = kBinsPerThread * t + k;
        // threshold bin: first bin with pre[b] < want <= pre[b+1].  Everything at or below it is handed to k_rank
        // grouped by bin (E.pre gives every bin its slice of the scratch array); the overshoot of the threshold
        // bin comes back to FRONT from there.
        if (pre[bin] < want && want <= pre[bin + 1]) s_bstar = (uint32_t)bin;
        // spill: FRONT grew past f_max -> keep the bins that hold the batch plus ~f_keep more
        if (rebased && n > E.f_max) {
            const uint32_t keepn = want + front_keep(E);
            if (pre[bin] < keepn && keepn <= pre[bin + 1]) s_spill = (uint32_t)bin;
        }
        // Between two rebase iterations the histogram is only maintained below `hbin`, the first bin under which FRONT
        // holds (kRefillPeriod + 2) batches now: the next kRefillPeriod pops take at most kRefillPeriod batches from the
        // bottom and pushes only add, so every threshold bin until the next recount lies below it — and k_commit need not
        // count the bulk of the children, which land higher (a global atomic per child and bin was a third of its time).
        if (rebased && pre[bin] < hwant && hwant <= pre[bin + 1]) s_hbin = (uint32_t)bin + 1u;
        E.pre[bin] = pre[bin];
        E.fill[bin] = 0;
    }
    for (int i = NBIN + t; i < kSegs; i += 1024) E.fill[i] = 0;  // (segments of a giant iteration)
    if (t == 1023) E.pre[NBIN] = pre[NBIN];
    __syncthreads();
    // A threshold bin too large for k_rank's LDS bucketing (massive cost ties: an integer-valued heuristic makes every
    // f-level one tie group of up to millions of entries) is not handed to k_rank whole: k_sel_collect refines it across
    // the grid first ("giant" iteration) and k_rank only sees the few thousand entries around the batch's end.
    const uint32_t giant_limit = g_tune[3] > 0 ? (uint32_t)g_tune[3] : kGiantBinDefault;
    const bool giant = E.coop && !*E.coop_off && want != 0 && pre[s_bstar + 1] - pre[s_bstar] > giant_limit;
    if (giant)
        for (int i = t; i < kMaxLevels * kSub; i += 1024) E.subhist[i] = 0;
    for (int k = 0; k < kBinsPerThread; k++) {
        const uint32_t bin = kBinsPerThread * t + k;
        if (bin <= s_bstar && want != 0) {
            const uint32_t cn = pre[bin + 1] - pre[bin];
            if (cn > 256) atomicMax(&s_maxbin, cn);
            if (cn > (uint32_t)kSortCap) atomicAdd(&s_giant, 1u);
            // work list of k_rank: one workgroup per bin of more than kTinyBin entries (smaller bins: a thread per entry);
            // the segments that replace a giant threshold bin are listed by k_sel_collect
            if (cn > kTinyBin && !(giant && bin == s_bstar)) {
                // a bin that fits k_rank's LDS path is shared between up to eight workgroups (about a thousand entries each)
                uint32_t G = rank_shares(cn);
                G = G > 8u ? 8u : G;
                const uint32_t at = atomicAdd(&s_nbig, G);
                for (uint32_t g = 0; g < G; g++) E.big_list[at + g] = bin | (g << 16) | (G << 20);
            }
        }
    }
    __syncthreads();  // (s_nbig, s_maxbin, s_giant are complete)
    if (t == 0) {
        if (rebased) {
            uint64_t fmn = ~0ull, fmx = 0;
            for (int w = 0; w < 16; w++) {
                fmn = s_red[0][w] < fmn ? s_red[0][w] : fmn;
                fmx = s_red[1][w] > fmx ? s_red[1][w] : fmx;
            }
            c->cur_f = cb;
            c->rng[cb].kmin = fmn;
            c->rng[cb].kmax = fmx;
            c->hbin = s_hbin;
        }
        if (refill) {
            if (compact) {  // the compacted copy becomes BACK
                c->cur_b ^= 1;
                c->back_dead.v = 0;
                c->compact = 0;
            }
            c->refill = 0;
        }
        c->want = want;
        c->bstar = s_bstar;
        c->tseg = s_bstar;
        c->pre_b = pre[s_bstar];
        c->cn_star = pre[s_bstar + 1] - pre[s_bstar];
        c->giant = giant ? 1u : 0u;
        c->gbar.v = 0;
        c->grange.kmin = ~0ull;
        c->grange.kmax = 0;
        c->grange.imin = ~0u;
        c->grange.imax = 0;
        c->sel_kmin = new_kmin;
        c->shift = new_shift;
        uint32_t sp = s_spill;
        if (sp < NBIN - 1) {
            const uint64_t top = new_kmin + (((uint64_t)sp + 1) << new_shift) - 1;
            if (top >= new_kmin && top < T0) c->T = top;  else sp = NBIN;
        } else {
            sp = NBIN;
        }
        if (rebased) c->front_above = sp < NBIN ? pre[NBIN] - pre[sp + 1] : 0u;
        c->spill_bin = sp;  // T was lowered to the top of this bin: the entries above it leave for BACK in the next rebase pass
        s_sp = sp;
        c->goal_best = ~0ull;
        c->first_solved = NIL;
        if (want == 0) {  // OPEN ran empty: no solution reachable
            c->failed = 1;
            c->done = 1;
        }
        c->n_big = s_nbig;
        c->n_ord = want ? pre[s_bstar + 1] : 0;
        // k_sel_collect tombstones what leaves FRONT: the bins handed to k_rank — which puts the overshoot of the threshold
        // bin back into the same slots, so only the batch itself stays dead
        c->ret_n.v = 0;
        c->dbg_maxsub = 0;
        c->front_dead.v = (rebased ? 0u : dead0) + want;
        c->dbg_nord = want ? pre[s_bstar + 1] : 0;
        c->dbg_maxbin = s_maxbin;
        c->dbg_giant = s_giant;
        c->dbg_giant_seen = giant_seen + s_giant;
    }
    __syncthreads();
    // the histogram of what stays in FRONT: bins at or below the threshold bin leave (the threshold bin's overshoot
    // comes back from k_rank: pre[bstar+1] - want entries), bins above the spill bin are no longer tracked
    {
        const uint32_t bstar = s_bstar, sp = s_sp;
        for (int k = 0; k < kBinsPerThread; k++) {
            const uint32_t bin = kBinsPerThread * t + k;
            if (want == 0) break;
            if (bin < bstar || bin > sp)
                E.hist[bin] = 0;
            else if (bin == bstar)
                E.hist[bin] = pre[bin + 1] - want;
        }
    }
}

// the (cost key, node id) composite: the total order of OPEN (astar.py:64-67: cost, then push count)
typedef unsigned __int128 u128;
__device__ __forceinline__ u128 comp_of(uint64_t key, uint32_t id) { return ((u128)key << 32) | (u128)(id & ID_MASK); }
__device__ __forceinline__ int clz128(u128 v) {
    uint64_t hi = (uint64_t)(v >> 64), lo = (uint64_t)v;
    return hi ? __clzll((long long)hi) : 64 + (lo ? __clzll((long long)lo) : 64);
}

// S3: take the batch's bins out of FRONT, in place.  Only the keys are read (8 bytes an entry); an entry at or below
// the threshold bin is stashed in LDS (key + position), tombstoned, and placed — with its id, gathered then — into the
// scratch array grouped by bin (k_rank orders each bin): one global atomic per (workgroup, bin).
//
// Giant iterations (c->giant: the threshold bin is a cost-tie group of tens of thousands to millions of entries).  Moving
// the whole bin to the scratch array and letting ONE workgroup of k_rank cut it down (what round 2 did: 1-15 ms per pop)
// is replaced by a radix selection on the (key, id) composite carried out by this launch's whole grid, in place:
//   A  range of the bin's keys and ids (one pass over FRONT, four global atomics per workgroup)         | grid barrier
//   B  per level: 2048 sub-bins over the current composite range, counted per workgroup in LDS and      | grid barrier
//      added to the level's global array; every workgroup then finds the sub-bin holding the batch's
//      last entry and, while that sub-bin is still larger than k_rank's LDS capacity, descends into it
//   D  collection: bins below the threshold bin as usual; of the threshold bin the part below the last level's range
//      (certainly in the batch: segment bstar) and the last level's sub-bins up to the threshold one (segments
//      bstar+1+s); everything else is left untouched in FRONT — not even tombstoned.
// Workgroup 0 publishes the segments' offsets, the threshold segment and k_rank's work units.  The grid barrier is a
// counter in the control block (agent-scope release before the arrival, relaxed polling, one acquire after; every spin
// bounded): the launch is sized to be fully resident (2 workgroups per CU; single-instance engines only).
constexpr int kStashG = 1536;   // the stash of a giant iteration's collection pass (tiles of 1024 entries there)
struct CollectLds {
    // exclusive prefix of the selection histogram: computed by EVERY workgroup from E.hist in the iterations that have no
    // k_sel_scan launch (all but the rebase iterations), copied from E.pre otherwise
    uint32_t pre[NBIN + 4];
    uint32_t lcnt[NBIN];  // entries stashed per segment, then the workgroup's slice base; giant: continues in g.lcnt_ext
    union {
        struct {
            uint64_t st_key[kStash];
            uint32_t st_idx[kStash];
            uint16_t st_f[kStash];
            uint16_t nzf[kStash];  // the segments this workgroup stashed entries of (each once)
        } n;
        struct {
            uint32_t lcnt_ext[kSegs - NBIN + 3];
            uint32_t subpre[kSub + 4];  // exclusive prefix of the last level's sub-bin counts
            uint64_t st_key[kStashG];   // (doubles as the sub-bin counters of the histogram passes)
            uint32_t st_idx[kStashG];
            uint16_t st_f[kStashG];
            uint16_t nzf[kStashG];
        } g;
    } u;
    uint64_t red64[2][4];
    uint32_t red32[2][4];
    uint32_t wsum[4];
    uint32_t st_n, nz_n, ok, tsub, nb, bstar;
};
static_assert(sizeof(CollectLds) <= 80 * 1024, "two workgroups of k_sel_collect must fit one CU's LDS");
static_assert(offsetof(CollectLds, u) == offsetof(CollectLds, lcnt) + sizeof(uint32_t) * NBIN, "lcnt must run on into lcnt_ext");
static_assert(sizeof(uint64_t) * kStashG >= sizeof(uint32_t) * kSub, "sub-bin counters alias the giant stash");

// what a pop selects, identical in every workgroup of the launch
struct SelParams {
    uint32_t b, n;        // FRONT buffer and its physical size
    uint64_t kmin;        // binning in force
    uint32_t shift;
    uint32_t want, bstar, pre_b, cn_star;
    bool giant;
};

// All threads of the workgroup.  -> BAR_PASS; BAR_FAIL = a later barrier timed out or another workgroup gave up (the search is
// failed); BAR_BROKEN (first barrier only) = the launch is not fully resident — some workgroup waited kBarrierTimeoutFirst for
// the others — and the giant path is abandoned BY CONSENSUS: the workgroup that gives up sets bit 31 of the arrival counter with
// a compare-and-swap that only succeeds while the count is still short of the target, so either every workgroup passes
// (the count reached the target before anybody gave up: arrivals never clear the bit, nobody can give up afterwards because the
// swap's expected value is stale) or every workgroup — those polling, and those that only become resident after the early ones
// left — sees the bit and takes the same fallback.  Nothing of the search has been modified before the first barrier.
enum { BAR_FAIL = 0, BAR_PASS = 1, BAR_BROKEN = 2 };
__device__ __forceinline__ int collect_grid_barrier(const Eng& E, Ctl* c, CollectLds& L, uint32_t target, bool first) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave's stores / atomics have left
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t v = __hip_atomic_fetch_add(&c->gbar.v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        int res = BAR_PASS;
        const unsigned long long t0 = wall_clock64();
        // (knob 10: the first barrier gives up at once — what a launch that is not fully resident does after 0.25 s; tests)
        const unsigned long long limit = first ? (g_tune[10] ? 0ull : kBarrierTimeoutFirst) : kBarrierTimeout;
        for (;;) {
            if (v & GBAR_BROKEN) {
                res = first ? BAR_BROKEN : BAR_FAIL;
                break;
            }
            if (v >= target) break;
            __builtin_amdgcn_s_sleep(4);
            if (__hip_atomic_load(&c->failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                res = BAR_FAIL;
                break;
            }
            if (wall_clock64() - t0 > limit) {
                if (!first) {
                    res = BAR_FAIL;
                    break;
                }
                uint32_t expect = v;
                if (__hip_atomic_compare_exchange_strong(&c->gbar.v, &expect, v | GBAR_BROKEN, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT)) {
                    res = BAR_BROKEN;
                    break;
                }
                v = expect;  // the count moved on: look again (it may be complete now)
                continue;
            }
            v = __hip_atomic_load(&c->gbar.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (res == BAR_FAIL) {  // a workgroup died or stalled mid-selection: refuse to continue with half a selection
            __hip_atomic_store(&c->failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&c->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (res == BAR_BROKEN) {
            __hip_atomic_store(E.coop_off, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        L.ok = (uint32_t)res;
    }
    __syncthreads();
    return (int)L.ok;
}

// -> BAR_PASS: the batch's segments are collected; BAR_FAIL: the search failed; BAR_BROKEN: the launch is not fully resident,
// nothing was touched — the caller collects the bins the ordinary way (k_rank then streams the giant bin on one workgroup)
__device__ __noinline__ int collect_giant(const Eng& E, Ctl* c, CollectLds& L, const SelParams P) {
    const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const uint32_t n = P.n;
    const uint64_t bkmin = P.kmin;
    const uint32_t shift = P.shift, bstar = P.bstar, want = P.want;
    const uint32_t pre_b = P.pre_b;
    uint64_t* __restrict__ keys = E.open_key[P.b];
    const uint32_t* __restrict__ ids = E.open_id[P.b];
    constexpr uint32_t ITEMS = 8, TILE = 256 * ITEMS;
    const uint32_t ntiles = (n + TILE - 1) / TILE;
    uint32_t phase = 0;
    // ---- A: key / id range of the threshold bin
    {
        uint64_t kmn = ~0ull, kmx = 0;
        uint32_t imn = ~0u, imx = 0;
        for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            uint64_t k[ITEMS];
            uint32_t id[ITEMS];
#pragma unroll
            for (uint32_t i = 0; i < ITEMS; i++) {
                const uint32_t idx = tile * TILE + i * 256 + t, ic = idx < n ? idx : n - 1;
                k[i] = keys[ic];
                id[i] = ids[ic] & ID_MASK;
                if (idx >= n) k[i] = DEAD;
            }
#pragma unroll
            for (uint32_t i = 0; i < ITEMS; i++) {
                const bool in = k[i] != DEAD && bin_of(k[i], bkmin, shift) == bstar;
                kmn = in && k[i] < kmn ? k[i] : kmn;
                kmx = in && k[i] > kmx ? k[i] : kmx;
                imn = in && id[i] < imn ? id[i] : imn;
                imx = in && id[i] > imx ? id[i] : imx;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const uint64_t a = __shfl_xor(kmn, o), z = __shfl_xor(kmx, o);
            const uint32_t x = __shfl_xor(imn, o), y = __shfl_xor(imx, o);
            kmn = a < kmn ? a : kmn;
            kmx = z > kmx ? z : kmx;
            imn = x < imn ? x : imn;
            imx = y > imx ? y : imx;
        }
        if (lane == 0) {
            L.red64[0][wv] = kmn;
            L.red64[1][wv] = kmx;
            L.red32[0][wv] = imn;
            L.red32[1][wv] = imx;
        }
        __syncthreads();
        if (t == 0) {
            for (int w = 1; w < 4; w++) {
                kmn = L.red64[0][w] < kmn ? L.red64[0][w] : kmn;
                kmx = L.red64[1][w] > kmx ? L.red64[1][w] : kmx;
                imn = L.red32[0][w] < imn ? L.red32[0][w] : imn;
                imx = L.red32[1][w] > imx ? L.red32[1][w] : imx;
            }
            if (kmn <= kmx) {  // (this workgroup saw entries of the bin)
                atomicMin((unsigned long long*)&c->grange.kmin, (unsigned long long)kmn);
                atomicMax((unsigned long long*)&c->grange.kmax, (unsigned long long)kmx);
                atomicMin(&c->grange.imin, imn);
                atomicMax(&c->grange.imax, imx);
            }
        }
    }
    if (const int r = collect_grid_barrier(E, c, L, ++phase * gridDim.x, true); r != BAR_PASS) return r;
    const uint64_t gkmin = __hip_atomic_load(&c->grange.kmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t gkmax = __hip_atomic_load(&c->grange.kmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t gimin = __hip_atomic_load(&c->grange.imin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t gimax = __hip_atomic_load(&c->grange.imax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- B: radix selection on the composite, level by level.  An entry's offset is comp(key, id) - V0, in [0, span].
    u128 V0 = ((u128)gkmin << 32) + (u128)gimin;
    u128 span = ((u128)(gkmax - gkmin) << 32) + (u128)(gimax - gimin);  // (upper bound: ids of the top key are <= imax)
    uint32_t need = want - pre_b, below = 0, shc = 0, tsub = 0;
    const uint32_t giant_limit = g_tune[3] > 0 ? (uint32_t)g_tune[3] : kGiantBinDefault;
    uint32_t* lh = reinterpret_cast<uint32_t*>(L.u.g.st_key);  // (the stash is idle until the collection pass)
    uint32_t* subpre = L.u.g.subpre;
    for (int lvl = 0;; lvl++) {
        const uint32_t bits = span ? (uint32_t)(128 - clz128(span)) : 0u;
        shc = bits > 11u ? bits - 11u : 0u;
        for (uint32_t i = t; i < (uint32_t)kSub; i += 256) lh[i] = 0;
        __syncthreads();
        for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            uint64_t k[ITEMS];
            uint32_t id[ITEMS];
#pragma unroll
            for (uint32_t i = 0; i < ITEMS; i++) {
                const uint32_t idx = tile * TILE + i * 256 + t, ic = idx < n ? idx : n - 1;
                k[i] = keys[ic];
                id[i] = ids[ic];
                if (idx >= n) k[i] = DEAD;
            }
#pragma unroll
            for (uint32_t i = 0; i < ITEMS; i++) {
                if (k[i] == DEAD || bin_of(k[i], bkmin, shift) != bstar) continue;
                const u128 v = comp_of(k[i], id[i]);
                if (v < V0 || v - V0 > span) continue;
                atomicAdd(&lh[(uint32_t)((v - V0) >> shc)], 1u);
            }
        }
        __syncthreads();
        uint32_t* gh = E.subhist + (size_t)lvl * kSub;
        for (uint32_t i = t; i < (uint32_t)kSub; i += 256)
            if (lh[i]) atomicAdd(&gh[i], lh[i]);
        if (collect_grid_barrier(E, c, L, ++phase * gridDim.x, false) != BAR_PASS) return BAR_FAIL;
        // every workgroup: prefix over the level's counts, the sub-bin holding the need-th entry
        constexpr int PER = kSub / 256;
        uint32_t v[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            v[k] = __hip_atomic_load(&gh[PER * t + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sum += v[k];
        }
        uint32_t incl = sum;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(incl, o);
            if (lane >= (uint32_t)o) incl += u;
        }
        if (lane == 63) L.wsum[wv] = incl;
        if (t == 0) L.tsub = ~0u;
        __syncthreads();
        uint32_t run = incl - sum;
        for (uint32_t w = 0; w < wv; w++) run += L.wsum[w];
#pragma unroll
        for (int k = 0; k < PER; k++) {
            subpre[PER * t + k] = run;
            if (run < need && need <= run + v[k]) L.tsub = PER * t + k;
            run += v[k];
        }
        if (t == 255) subpre[kSub] = run;
        __syncthreads();
        tsub = L.tsub;
        if (tsub == ~0u) {  // counts and FRONT disagree (cannot happen)
            if (t == 0) c->failed = 1, c->done = 1;
            return BAR_FAIL;
        }
        const uint32_t cn = subpre[tsub + 1] - subpre[tsub];
        if (cn <= giant_limit || shc == 0 || lvl + 1 >= kMaxLevels) break;
        // descend into the threshold sub-bin: everything below it is certainly in the batch
        below += subpre[tsub];
        need -= subpre[tsub];
        V0 += (u128)tsub << shc;
        span = ((u128)1 << shc) - 1;
        __syncthreads();  // (subpre / tsub are rewritten by the next level)
    }
    // ---- publish (workgroup 0): segment offsets, threshold segment, k_rank's work units
    const uint32_t seg0 = bstar + 1u;  // segment of the last level's sub-bin 0; segment bstar = the part below V0
    if (blockIdx.x == 0) {
        if (t == 0) {
            L.nb = c->n_big;
            E.pre[bstar + 1] = pre_b + below;
            c->tseg = seg0 + tsub;
            c->n_ord = pre_b + below + subpre[tsub + 1];
            c->dbg_nord = pre_b + below + subpre[tsub + 1];
        }
        __syncthreads();
        for (uint32_t sg = t; sg <= tsub + 1u; sg += 256) {  // sg 0 = the below part, sg 1 + s = sub-bin s
            const uint32_t cn = sg == 0 ? below : subpre[sg] - subpre[sg - 1];
            if (sg > 0) E.pre[seg0 + sg] = pre_b + below + subpre[sg];
            if (cn > kTinyBin) {
                uint32_t G = rank_shares(cn);
                G = G > 8u ? 8u : G;
                const uint32_t at = atomicAdd(&L.nb, G);
                for (uint32_t g = 0; g < G; g++) E.big_list[at + g] = (bstar + sg) | (g << 16) | (G << 20);
            }
        }
        __syncthreads();
        if (t == 0) c->n_big = L.nb;
    }
    // ---- D: collection (tiles of 1024 entries: the stash is smaller here)
    uint32_t* lcnt = L.lcnt;  // (runs on into u.g.lcnt_ext: kSegs entries)
    for (uint32_t i = t; i < (uint32_t)(kSegs + 3); i += 256) lcnt[i] = 0;
    if (t == 0) {
        L.st_n = 0;
        L.nz_n = 0;
    }
    __syncthreads();
    uint64_t* st_key = L.u.g.st_key;
    uint32_t* st_idx = L.u.g.st_idx;
    uint16_t *st_f = L.u.g.st_f, *nzf = L.u.g.nzf;
    auto seg_base = [&](uint32_t f) -> uint32_t {
        return f < bstar ? L.pre[f] : f == bstar ? pre_b : pre_b + below + subpre[f - seg0];
    };
    auto seg_end = [&](uint32_t f) -> uint32_t {
        return f < bstar ? L.pre[f + 1] : f == bstar ? pre_b + below : pre_b + below + subpre[f - seg0 + 1];
    };
    auto flush = [&]() {
        const uint32_t ns = L.st_n < (uint32_t)kStashG ? L.st_n : (uint32_t)kStashG;
        const uint32_t nz = L.nz_n;
        for (uint32_t i = t; i < nz; i += 256) {
            const uint32_t f = nzf[i];
            lcnt[f] = atomicAdd(&E.fill[f], lcnt[f]);
        }
        __syncthreads();
        for (uint32_t p = t; p < ns; p += 256) {
            const uint32_t f = st_f[p];
            const uint32_t id = ids[st_idx[p]];
            const uint32_t pos = seg_base(f) + atomicAdd(&lcnt[f], 1u);
            if (pos < seg_end(f)) {
                E.tmp_key[pos] = st_key[p];
                E.tmp_id[pos] = id;
                E.tmp_f[pos] = (uint16_t)f;
                E.tmp_idx[pos] = st_idx[p];
            } else {
                c->failed = 1;  // counts and FRONT disagree (cannot happen): refuse to write outside the segment's slice
            }
        }
        __syncthreads();
        for (uint32_t i = t; i < nz; i += 256) lcnt[nzf[i]] = 0;
        __syncthreads();
        if (t == 0) {
            L.st_n = 0;
            L.nz_n = 0;
        }
        __syncthreads();
    };
    constexpr uint32_t ITEMS_D = 4, TILE_D = 256 * ITEMS_D;
    const uint32_t ntiles_d = (n + TILE_D - 1) / TILE_D;
    // (the same entries as in the passes above: workgroup w owned tiles w, w + grid, ... of 2048 entries = tile pairs here)
    for (uint32_t tile2 = blockIdx.x; tile2 < ntiles; tile2 += gridDim.x) {
        for (uint32_t half = 0; half < 2; half++) {
            const uint32_t tile = tile2 * 2 + half;
            if (tile < ntiles_d) {  // (uniform)
                uint64_t k[ITEMS_D];
                uint32_t id[ITEMS_D];
#pragma unroll
                for (uint32_t i = 0; i < ITEMS_D; i++) {
                    const uint32_t idx = tile * TILE_D + i * 256 + t, ic = idx < n ? idx : n - 1;
                    k[i] = keys[ic];
                    id[i] = ids[ic];
                    if (idx >= n) k[i] = DEAD;
                }
#pragma unroll
                for (uint32_t i = 0; i < ITEMS_D; i++) {
                    if (k[i] == DEAD) continue;
                    uint32_t f = bin_of(k[i], bkmin, shift);
                    if (f > bstar) continue;
                    if (f == bstar) {
                        const u128 v = comp_of(k[i], id[i]);
                        if (v >= V0) {
                            if (v - V0 > span) continue;
                            const uint32_t sub = (uint32_t)((v - V0) >> shc);
                            if (sub > tsub) continue;  // stays in FRONT, untouched
                            f = seg0 + sub;
                        }
                    }
                    const uint32_t idx = tile * TILE_D + i * 256 + t;
                    const uint32_t p = atomicAdd(&L.st_n, 1u);
                    keys[idx] = DEAD;
                    if (p < (uint32_t)kStashG) {
                        st_key[p] = k[i];
                        st_idx[p] = idx;
                        st_f[p] = (uint16_t)f;
                        if (atomicAdd(&lcnt[f], 1u) == 0u) nzf[atomicAdd(&L.nz_n, 1u)] = (uint16_t)f;
                    } else {  // stash full (cannot happen: it is flushed while a whole tile still fits): place directly
                        const uint32_t pos = seg_base(f) + atomicAdd(&E.fill[f], 1u);
                        if (pos < seg_end(f)) {
                            E.tmp_key[pos] = k[i];
                            E.tmp_id[pos] = id[i];
                            E.tmp_f[pos] = (uint16_t)f;
                            E.tmp_idx[pos] = idx;
                        } else {
                            c->failed = 1;
                        }
                    }
                }
            }
            __syncthreads();
            if (L.st_n + TILE_D > (uint32_t)kStashG) flush();
        }
    }
    __syncthreads();
    flush();
    return BAR_PASS;
}

// FUSED: this launch opens the iteration — there is no k_sel_scan in front of it.  Every workgroup derives the pop's
// geometry itself from the selection histogram (16 KB from L2 and a 4096-bin prefix in LDS: cheaper than a single-workgroup
// launch plus its boundary, which was 7 % of the iteration); workgroup 0 also records it for k_rank.  `want` was left by the
// previous iteration's k_commit (commit_ticket), the histogram's writeback is done by k_rank (nobody may change it while the
// workgroups of this launch read it).  Rebase iterations keep k_sel_scan (FUSED = false): it also takes FRONT's new key
// range, the spill decision and the refill's bookkeeping.
template <bool FUSED>
__global__ __launch_bounds__(256) void k_sel_collect(const Eng* __restrict__ engs) {
    const Eng& E = engs[blockIdx.y];
    Ctl* c = E.ctl;
    if (c->done) return;
    Stamp stamp(E, P_SEL_COLLECT);
    extern __shared__ __attribute__((aligned(16))) uint8_t collect_lds[];
    CollectLds& L = *reinterpret_cast<CollectLds*>(collect_lds);
    const uint32_t t = threadIdx.x;
    SelParams P;
    {
        // (everything requested together: a read that waits for another is a trip to memory)
        const uint32_t b = c->cur_f, on0 = c->open_n[0].v, on1 = c->open_n[1].v;
        P.b = b;
        P.n = b ? on1 : on0;
        P.kmin = c->sel_kmin;
        P.shift = c->shift;
        P.want = c->want;
    }
    if constexpr (FUSED) {
        constexpr int PER = NBIN / 256;  // 16 bins per thread
        const uint32_t lane = t & 63, wv = t >> 6;
        uint32_t v[PER], sum = 0;
        {
            const uint4* h4 = reinterpret_cast<const uint4*>(E.hist + PER * t);
#pragma unroll
            for (int q = 0; q < PER / 4; q++) {
                const uint4 x = h4[q];
                v[4 * q] = x.x;
                v[4 * q + 1] = x.y;
                v[4 * q + 2] = x.z;
                v[4 * q + 3] = x.w;
            }
#pragma unroll
            for (int k = 0; k < PER; k++) sum += v[k];
        }
        uint32_t incl = sum;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(incl, o);
            if (lane >= (uint32_t)o) incl += u;
        }
        if (lane == 63) L.wsum[wv] = incl;
        if (t == 0) L.bstar = 0;
        __syncthreads();
        uint32_t run = incl - sum;
        for (uint32_t w = 0; w < wv; w++) run += L.wsum[w];
        const uint32_t want = P.want;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            L.pre[PER * t + k] = run;
            if (run < want && want <= run + v[k]) L.bstar = PER * t + k;
            run += v[k];
        }
        if (t == 255) L.pre[NBIN] = run;
        __syncthreads();
        if (want == 0) {  // OPEN ran empty: no solution reachable
            if (blockIdx.x == 0 && t == 0) {
                c->failed = 1;
                c->done = 1;
            }
            return;
        }
        P.bstar = L.bstar;
        P.pre_b = L.pre[P.bstar];
        P.cn_star = L.pre[P.bstar + 1] - P.pre_b;
        const uint32_t giant_limit = g_tune[3] > 0 ? (uint32_t)g_tune[3] : kGiantBinDefault;
        P.giant = E.coop && !*E.coop_off && P.cn_star > giant_limit;
        if (blockIdx.x == 0) {
            // the record k_rank (and the rest of the iteration) reads — what k_sel_scan writes in a rebase iteration
            for (uint32_t i = t; i <= (uint32_t)NBIN; i += 256) E.pre[i] = L.pre[i];
            if (t == 0) {
                L.nb = 0;
                L.st_n = 0;  // (borrowed: largest bin / bins beyond the LDS sort capacity, diagnostics)
                L.nz_n = 0;
            }
            __syncthreads();
            for (uint32_t bin = t; bin <= P.bstar; bin += 256) {
                const uint32_t cn = L.pre[bin + 1] - L.pre[bin];
                if (cn > 256) atomicMax(&L.st_n, cn);
                if (cn > (uint32_t)kSortCap) atomicAdd(&L.nz_n, 1u);
                if (cn > kTinyBin && !(P.giant && bin == P.bstar)) {
                    uint32_t G = rank_shares(cn);
                    G = G > 8u ? 8u : G;
                    const uint32_t at = atomicAdd(&L.nb, G);
                    for (uint32_t g = 0; g < G; g++) E.big_list[at + g] = bin | (g << 16) | (G << 20);
                }
            }
            __syncthreads();
            if (t == 0) {
                c->bstar = P.bstar;
                c->tseg = P.bstar;
                c->giant = P.giant ? 1u : 0u;
                c->pre_b = P.pre_b;
                c->cn_star = P.cn_star;
                c->spill_bin = NBIN;
                c->n_big = L.nb;
                c->n_ord = L.pre[P.bstar + 1];
                c->front_dead.v += want;  // k_sel_collect tombstones what leaves FRONT; k_rank puts the overshoot back
                c->dbg_nord = L.pre[P.bstar + 1];
                c->dbg_maxbin = L.st_n;
                c->dbg_giant = L.nz_n;
                c->dbg_giant_seen += L.nz_n;
                c->dbg_maxsub = 0;
            }
            __syncthreads();
        }
    } else {
        P.bstar = c->bstar;
        P.giant = c->giant != 0;
        P.pre_b = c->pre_b;
        P.cn_star = c->cn_star;
        if (P.want == 0) return;
        for (uint32_t i = t; i <= P.bstar + 1u; i += 256) L.pre[i] = E.pre[i];
        __syncthreads();
    }
    if (P.giant) {
        if (collect_giant(E, c, L, P) != BAR_BROKEN) return;
        // Not every workgroup of this launch is resident (the GPU is shared): every workgroup arrives here — nothing has been
        // touched yet — and collects the bins the ordinary way; workgroup 0 takes back what was published for k_rank: the
        // threshold bin is an ordinary (large) bin again, streamed by one workgroup there.
        if (blockIdx.x == 0 && t == 0) {
            const uint32_t cn = P.cn_star;
            uint32_t G = rank_shares(cn);
            G = G > 8u ? 8u : G;
            const uint32_t at = c->n_big;
            if (cn > kTinyBin) {
                for (uint32_t g = 0; g < G; g++) E.big_list[at + g] = P.bstar | (g << 16) | (G << 20);
                c->n_big = at + G;
            }
            c->giant = 0;
        }
        P.giant = false;
        __syncthreads();
    }
    const uint32_t n = P.n;
    const uint64_t kmin = P.kmin;
    const uint32_t shift = P.shift, bstar = P.bstar;
    uint64_t* __restrict__ keys = E.open_key[P.b];
    const uint32_t* __restrict__ ids = E.open_id[P.b];
    constexpr uint32_t ITEMS = 8, TILE = 256 * ITEMS;
    const uint32_t ntiles = (n + TILE - 1) / TILE;
    uint64_t* st_key = L.u.n.st_key;
    uint32_t* st_idx = L.u.n.st_idx;
    uint16_t *st_f = L.u.n.st_f, *nzf = L.u.n.nzf;
    for (int i = t; i < NBIN; i += 256) L.lcnt[i] = 0;
    if (t == 0) {
        L.st_n = 0;
        L.nz_n = 0;
    }
    __syncthreads();
    // The stash goes out (workgroup-collective, call after a barrier): one global atomic per bin this workgroup touched
    // reserves its slice of the bin, then every stashed entry takes its slot inside the slice (LDS atomic).  (Walking all
    // bins up to the threshold bin instead — a returning global atomic, and a wait, per non-empty bin and loop round — was
    // most of this kernel's time; the list of touched bins is a few dozen long: one round.)
    auto flush = [&]() {
        const uint32_t ns = L.st_n < (uint32_t)kStash ? L.st_n : (uint32_t)kStash;
        const uint32_t nz = L.nz_n;
        for (uint32_t i = t; i < nz; i += 256) {
            const uint32_t f = nzf[i];
            L.lcnt[f] = atomicAdd(&E.fill[f], L.lcnt[f]);
        }
        __syncthreads();
        for (uint32_t p = t; p < ns; p += 256) {
            const uint32_t f = st_f[p];
            const uint32_t id = ids[st_idx[p]];
            const uint32_t pos = L.pre[f] + atomicAdd(&L.lcnt[f], 1u);
            if (pos < L.pre[f + 1]) {
                E.tmp_key[pos] = st_key[p];
                E.tmp_id[pos] = id;
                E.tmp_f[pos] = (uint16_t)f;
                E.tmp_idx[pos] = st_idx[p];
            } else {
                c->failed = 1;  // histogram and FRONT disagree (cannot happen): refuse to write outside the bin's slice
            }
        }
        __syncthreads();
        for (uint32_t i = t; i < nz; i += 256) L.lcnt[nzf[i]] = 0;
        __syncthreads();
        if (t == 0) {
            L.st_n = 0;
            L.nz_n = 0;
        }
        __syncthreads();
    };
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint64_t k[ITEMS];
        uint32_t dest = 0;  // 2 bits per item: 0 stays (or dead), 1 scratch (ordered by k_rank)
        // all loads of the tile first (unconditional, index-clamped: they stay in flight together) ...
#pragma unroll
        for (uint32_t i = 0; i < ITEMS; i++) {
            const uint32_t idx = tile * TILE + i * 256 + t, ic = idx < n ? idx : n - 1;
            k[i] = keys[ic];
            if (idx >= n) k[i] = DEAD;
        }
#pragma unroll
        for (uint32_t i = 0; i < ITEMS; i++) {
            const uint32_t f = bin_of(k[i], kmin, shift);
            const uint32_t d = (k[i] != DEAD && f <= bstar) ? 1u : 0u;
            dest |= d << (2 * i);
        }
        // ... then the few entries bound for the scratch array (about one in seventy).  (Wave-aggregating the two LDS
        // atomics — one reservation per wave, one count per group of equal bins — measured slower: 17.5 vs 13.6 us.)
        if (dest & 0x5555u & ~(dest >> 1)) {
#pragma unroll
            for (uint32_t i = 0; i < ITEMS; i++) {
                if (((dest >> (2 * i)) & 3u) != 1u) continue;
                const uint32_t idx = tile * TILE + i * 256 + t;
                const uint32_t f = bin_of(k[i], kmin, shift);
                const uint32_t p = atomicAdd(&L.st_n, 1u);
                keys[idx] = DEAD;
                if (p < kStash) {
                    st_key[p] = k[i];
                    st_idx[p] = idx;
                    st_f[p] = (uint16_t)f;
                    if (atomicAdd(&L.lcnt[f], 1u) == 0u) nzf[atomicAdd(&L.nz_n, 1u)] = (uint16_t)f;
                } else {  // stash full (cannot happen: it is flushed while a whole tile still fits): place directly
                    const uint32_t pos = L.pre[f] + atomicAdd(&E.fill[f], 1u);
                    if (pos < L.pre[f + 1]) {
                        E.tmp_key[pos] = k[i];
                        E.tmp_id[pos] = ids[idx];
                        E.tmp_f[pos] = (uint16_t)f;
                        E.tmp_idx[pos] = idx;
                    } else {
                        c->failed = 1;  // histogram and FRONT disagree (cannot happen): refuse to write outside the bin's slice
                    }
                }
            }
        }
        // a tile can add up to TILE entries: make room when fewer are left (only ever true when a bin holds a large share
        // of FRONT — massive cost ties — where placing the overflow entry by entry meant millions of atomics on ONE counter)
        __syncthreads();
        if (L.st_n + TILE > (uint32_t)kStash) flush();
    }
    __syncthreads();
    flush();
}

// ---------------------------------------------------------------------------------------------
// k_rank: exact order of the batch.  The scratch array holds every entry at or below the threshold bin, grouped by bin.
//   * bins of at most kTinyBin entries (most bins, about half the entries): one THREAD per entry, the bins' slice of the
//     scratch array staged in LDS, rank = entries in lower bins + smaller (key,id) pairs inside the bin;
//   * larger bins are work units of k_sel_scan's list, one per workgroup (a bin of more than 1024 entries is shared by up
//     to eight workgroups).  Up to kLdsEnt entries the bin is read ONCE into registers and bucketed + ranked in LDS:
//       1  smallest key / id and their spans                          2  counts of up to 2048 sub-bins, prefix, the sub-bin
//       3  scatter of the sub-bins the batch needs into LDS;             holding the last entry the batch still needs
//          the rest of a threshold bin goes back to FRONT             4  rank inside the (few-entry) sub-bin = pop rank
//     a sub-bin that is still large (ids of a tie group clustered, a few distinct keys with huge multiplicities) becomes a
//     work item of its own, reading the slice the scatter wrote to the second scratch array (ping-pong);
//   * bins beyond kLdsEnt entries (a whole OPEN tied on one cost) stream through the same four passes from HBM on the
//     96-bit (key,id) composite, only counters in LDS: any size, one workgroup.
// Ranks below `want` are the batch in pop order (which fixes the children's node ids and therefore every later tie-break);
// the overshoot of the threshold bin returns to the FRONT slots k_sel_collect took it from.
// ---------------------------------------------------------------------------------------------

constexpr int kRankStack = 512;    // pending oversized sub-bins of one bin
constexpr uint32_t kDirectMax = 512;  // items up to this size are ranked all-pairs out of LDS
constexpr uint32_t kSubMaxDefault = 512;  // sub-bins up to this size are ranked in place, larger ones are refined again
#define kSubMax ((uint32_t)(g_tune[1] ? g_tune[1] : (int)kSubMaxDefault))

struct RankItem {
    uint32_t off, n, need, src;  // slice [off, off+n) of scratch array `src` (0 tmp, 1 ord); pop rank of its first entry = off
    uint32_t g, G;               // this workgroup does share g of G of the item (a large bin is split between workgroups)
    // Refining a sub-bin ping-pongs between the two scratch arrays; the second hop writes into the FIRST array — which
    // the other workgroups of a shared bin may still be reading.  Descendants of a shared item therefore never refine
    // again (an oversized sub-bin of theirs is ranked in place: slow, rare, exact).
    uint32_t nopush;
};
struct RankShared {
    uint32_t cnt[kSub + 64];  // sub-bin counts, then running scatter slots (+ one idle word per lane, see the LDS path)
    uint32_t off[kSub + 1];  // exclusive prefix
    uint32_t wsum[16];
    uint64_t red_lo[32], red_hi[32];
    uint64_t vmin_hi, vmin_lo;
    uint64_t kmin, kspan;  // LDS path: the item's smallest key and key span,
    uint32_t imin, ispan;  // and the same for its (masked) ids
    uint32_t bits, tsub, sp, fail;
    uint32_t ret_base, ret_cnt;            // one reservation of return numbers per work item for the entries it hands back
    RankItem stack[kRankStack];
};

// one entry of the batch at its pop rank
__device__ __forceinline__ void emit_pop(const Eng& E, Ctl* c, uint32_t rank, uint64_t key, uint32_t id) {
    E.pop_key[rank] = key;
    E.pop_id[rank] = id;  // flag included: k_expand masks it (and writes the parents' path costs, pop_g)
    if (id & ID_SOLVED) {  // rare: a goal among the popped — nothing else in the pop reads the node pool
        if (E.sem == DCA_SEM_PY)
            atomicMin(&c->goal_best, ((unsigned long long)(uint32_t)E.g[id & ID_MASK] << 32) | rank);
        else
            atomicMin(&c->first_solved, rank);
    }
}
// What k_rank does not pop (the overshoot of the threshold bin) goes back into FRONT — into the slots k_sel_collect
// tombstoned: return number r (a global counter, at most n_ord - want of them) takes the slot scratch entry r came from
// (E.tmp_idx[r]; any one-to-one assignment will do).  FRONT's physical size and key range do not change.
// One ranked entry: into the batch (pop order) or back to FRONT.  Wave-collective (one returning global atomic per
// wave) — fine once per thread (the small-bin pass), NOT inside a loop: the large-bin path uses ret_put.
__device__ __forceinline__ void emit_ranked(const Eng& E, Ctl* c, uint32_t nf, bool live, uint32_t rank, uint32_t want,
                                            uint64_t key, uint32_t id) {
    if (live && rank < want) emit_pop(E, c, rank, key, id);
    const bool back = live && rank >= want;
    const unsigned long long mask = __ballot(back);
    if (mask == 0) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t basep = 0;
    if (lane == leader) basep = atomicAdd(&c->ret_n.v, (uint32_t)__popcll(mask));
    basep = __shfl(basep, leader);
    if (back) {
        const uint32_t slot = E.tmp_idx[basep + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))];
        E.open_key[nf][slot] = key;
        E.open_id[nf][slot] = id;
    }
}
// The entries a work item hands back take return numbers of ONE reservation made for the whole item; their positions
// inside it come from an LDS counter.  (A returning global atomic per wave and loop round, sixteen rounds deep, was what
// made this path take 20-45 us.)
__device__ __forceinline__ void ret_begin(Ctl* c, RankShared& S, uint32_t nf, uint32_t count) {
    __syncthreads();
    if (threadIdx.x == 0) {
        S.ret_base = count ? atomicAdd(&c->ret_n.v, count) : 0u;
        S.ret_cnt = 0;
    }
    __syncthreads();
}
struct RetAcc {};  // (nothing to accumulate: an entry that goes back was inside FRONT's key range before)
// wave-collective (call from wave-uniform control flow): the lanes with `pred` take consecutive return numbers, ONE LDS
// atomic per wave and call.  (Per-entry atomics on the shared counter — all on one LDS address — cost 30-60 us.)
__device__ __forceinline__ void ret_put(const Eng& E, Ctl* c, RankShared& S, uint32_t nf, bool pred, uint64_t key,
                                        uint32_t id, RetAcc& acc) {
    const unsigned long long mask = __ballot(pred);
    if (mask == 0) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t basep = 0;
    if (lane == leader) basep = atomicAdd(&S.ret_cnt, (uint32_t)__popcll(mask));
    basep = __shfl(basep, leader);
    if (pred) {
        const uint32_t slot = E.tmp_idx[S.ret_base + basep + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))];
        E.open_key[nf][slot] = key;
        E.open_id[nf][slot] = id;
    }
}
__device__ __forceinline__ void ret_end(Ctl* c, uint32_t nf, RetAcc acc) {}
// NJ entries of one thread handed back at once (bit j of `preds`): one LDS atomic per wave for all of them, their slot
// numbers fetched with unconditional (index-clamped) loads — all in flight together — then predicated stores.  A
// ret_put per entry made every round wait for its own E.tmp_idx round trip.  Wave-collective.
template <int NJ>
__device__ __forceinline__ void ret_put_many(const Eng& E, RankShared& S, uint32_t nf, uint32_t n_ord, uint32_t preds,
                                             const uint64_t* key, const uint32_t* id) {
    const uint32_t cnt = (uint32_t)__popc(preds);
    const int lane = threadIdx.x & 63;
    uint32_t incl = cnt;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(incl, o);
        if (lane >= o) incl += u;
    }
    const uint32_t total = __shfl(incl, 63);
    if (total == 0) return;
    uint32_t basep = 0;
    if (lane == 63) basep = atomicAdd(&S.ret_cnt, total);
    basep = __shfl(basep, 63);
    uint32_t r = S.ret_base + basep + incl - cnt;
    uint32_t slot[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        slot[j] = E.tmp_idx[r < n_ord ? r : n_ord - 1];
        r += (preds >> j) & 1u;
    }
#pragma unroll
    for (int j = 0; j < NJ; j++)
        if ((preds >> j) & 1u) {
            E.open_key[nf][slot[j]] = key[j];
            E.open_id[nf][slot[j]] = id[j];
        }
}

// Entries of the bins of at most kTinyBin entries, a thread per entry: rank = entries in lower bins + smaller composites
// inside the bin.  A workgroup takes RT consecutive scratch positions; the bins they belong to form one contiguous
// range of the scratch array (it is grouped by bin), which is staged into LDS once — together with the bins' offsets —
// so the all-pairs loops run out of LDS.  Falls back to global loads when the range does not fit.
__device__ __forceinline__ void rank_small_entries(const Eng& E, Ctl* c, RankShared& S, uint64_t* LK, uint32_t* LI,
                                                   uint32_t nf, uint32_t p0, uint32_t n_ord, uint32_t want) {
    const uint32_t t = threadIdx.x;
    const uint32_t plast = (p0 + RT - 1 < n_ord ? p0 + RT - 1 : n_ord - 1);
    const uint32_t f_lo = E.tmp_f[p0], f_hi = E.tmp_f[plast];
    const uint32_t lo = E.pre[f_lo], hi = E.pre[f_hi + 1];
    const bool staged = hi - lo <= kLdsEnt && f_hi - f_lo + 2 <= (uint32_t)kSub;
    if (staged) {
        for (uint32_t i = t; i < hi - lo; i += RT) {
            LK[i] = E.tmp_key[lo + i];
            LI[i] = E.tmp_id[lo + i];
        }
        for (uint32_t i = t; i < f_hi - f_lo + 2; i += RT) S.off[i] = E.pre[f_lo + i];
    }
    __syncthreads();
    const uint32_t p = p0 + t;
    bool live = p < n_ord;
    uint64_t k = 0;
    uint32_t id = 0, rank = 0;
    if (live) {
        const uint32_t f = E.tmp_f[p];
        const uint32_t o = staged ? S.off[f - f_lo] : E.pre[f], e = staged ? S.off[f - f_lo + 1] : E.pre[f + 1];
        if (e - o > kTinyBin) {
            live = false;  // a large bin: ranked by its workgroup
        } else if (staged) {
            k = LK[p - lo];
            id = LI[p - lo];
            rank = o;
            for (uint32_t j = o - lo; j < e - lo; j++) rank += pair_less(LK[j], LI[j], k, id) ? 1u : 0u;
        } else {
            k = E.tmp_key[p];
            id = E.tmp_id[p];
            rank = o;
            for (uint32_t j = o; j < e; j++) rank += pair_less(E.tmp_key[j], E.tmp_id[j], k, id) ? 1u : 0u;
        }
    }
    emit_ranked(E, c, nf, live, rank, want, k, id);
    __syncthreads();
}

__device__ __forceinline__ uint32_t sub_of(uint64_t k, uint32_t id, u128 vmin, uint32_t shc, uint32_t nsub) {
    const u128 q = (comp_of(k, id) - vmin) >> shc;
    return q < (u128)nsub ? (uint32_t)q : nsub - 1u;
}


// exact composite range of the item from per-thread partial min / max -> S.vmin_*, S.bits; also clears the counters
__device__ __forceinline__ void rank_range(RankShared& S, u128 vmin, u128 vmax) {
    const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
    for (int o = 32; o > 0; o >>= 1) {
        uint64_t ahi = __shfl_xor((uint64_t)(vmin >> 64), o), alo = __shfl_xor((uint64_t)vmin, o);
        uint64_t bhi = __shfl_xor((uint64_t)(vmax >> 64), o), blo = __shfl_xor((uint64_t)vmax, o);
        u128 a = ((u128)ahi << 64) | alo, b = ((u128)bhi << 64) | blo;
        vmin = a < vmin ? a : vmin;
        vmax = b > vmax ? b : vmax;
    }
    if (lane == 0) {
        S.red_hi[wv] = (uint64_t)(vmin >> 64);
        S.red_lo[wv] = (uint64_t)vmin;
        S.red_hi[16 + wv] = (uint64_t)(vmax >> 64);
        S.red_lo[16 + wv] = (uint64_t)vmax;
    }
    for (uint32_t i = t; i < (uint32_t)kSub; i += RT) S.cnt[i] = 0;
    __syncthreads();
    if (t == 0) {
        u128 mn = ~(u128)0, mx = 0;
        for (int k = 0; k < RT / 64; k++) {
            u128 a = ((u128)S.red_hi[k] << 64) | S.red_lo[k], b = ((u128)S.red_hi[16 + k] << 64) | S.red_lo[16 + k];
            mn = a < mn ? a : mn;
            mx = b > mx ? b : mx;
        }
        const u128 range = mx - mn;
        S.bits = range ? (uint32_t)(128 - clz128(range)) : 0u;
        S.vmin_hi = (uint64_t)(mn >> 64);
        S.vmin_lo = (uint64_t)mn;
    }
    __syncthreads();
}

// the LDS path's range: smallest key / id and their spans from per-thread partials -> S.kmin, S.kspan, S.imin, S.ispan;
// also clears the counters
__device__ __forceinline__ void rank_range64(RankShared& S, uint64_t kmn, uint64_t kmx, uint32_t imn, uint32_t imx) {
    const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t a = __shfl_xor(kmn, o), b = __shfl_xor(kmx, o);
        const uint32_t x = __shfl_xor(imn, o), y = __shfl_xor(imx, o);
        kmn = a < kmn ? a : kmn;
        kmx = b > kmx ? b : kmx;
        imn = x < imn ? x : imn;
        imx = y > imx ? y : imx;
    }
    if (lane == 0) {
        S.red_lo[wv] = kmn;
        S.red_lo[16 + wv] = kmx;
        S.red_hi[wv] = imn;
        S.red_hi[16 + wv] = imx;
    }
    for (uint32_t i = t; i < (uint32_t)kSub; i += RT) S.cnt[i] = 0;
    __syncthreads();
    if (t == 0) {
        uint64_t a = ~0ull, b = 0, x = ~0ull, y = 0;
        for (int k = 0; k < RT / 64; k++) {
            a = S.red_lo[k] < a ? S.red_lo[k] : a;
            b = S.red_lo[16 + k] > b ? S.red_lo[16 + k] : b;
            x = S.red_hi[k] < x ? S.red_hi[k] : x;
            y = S.red_hi[16 + k] > y ? S.red_hi[16 + k] : y;
        }
        S.kmin = a;
        S.kspan = b - a;
        S.imin = (uint32_t)x;
        S.ispan = (uint32_t)(y - x);
    }
    __syncthreads();
}

// exclusive prefix of cnt[0..nsub) into off[0..nsub] (2 per thread), the sub-bin holding the need-th entry -> S.tsub,
// counters back to zero (they become the running scatter slots)
__device__ __forceinline__ void rank_prefix(RankShared& S, uint32_t nsub, uint32_t need) {
    constexpr int PER = kSub / RT;  // counters per thread (4)
    const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
    uint32_t v[PER], s4 = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        v[k] = PER * t + k < nsub ? S.cnt[PER * t + k] : 0u;
        s4 += v[k];
    }
    uint32_t incl = s4;
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t u = __shfl_up(incl, o);
        if (lane >= (uint32_t)o) incl += u;
    }
    if (lane == 63) S.wsum[wv] = incl;
    __syncthreads();
    uint32_t run = incl - s4;
    for (uint32_t w = 0; w < wv; w++) run += S.wsum[w];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const uint32_t i = PER * t + k;
        if (i < nsub) {
            S.off[i] = run;
            S.cnt[i] = 0;
            if (run < need && need <= run + v[k]) S.tsub = i;
            if (i + 1 == nsub) S.off[nsub] = run + v[k];
        }
        run += v[k];
    }
    __syncthreads();
}

// oversized sub-bins of the item just scattered become work items of their own (slice of the OTHER scratch array)
__device__ __forceinline__ void rank_push(RankShared& S, const RankItem& it, uint32_t tsub, uint32_t need, bool refinable,
                                          uint32_t lo = 0u, uint32_t hi = ~0u) {
    for (uint32_t sb = lo + threadIdx.x; sb <= tsub && sb < hi; sb += RT) {
        const uint32_t s0 = S.off[sb], e0 = S.off[sb + 1];
        if (e0 - s0 > kSubMax && refinable) {
            const uint32_t slot = atomicAdd(&S.sp, 1u);
            if (slot < (uint32_t)kRankStack)
                S.stack[slot] = RankItem{it.off + s0, e0 - s0, sb == tsub ? need - s0 : e0 - s0, it.src ^ 1u, 0u, 1u,
                                         (it.G > 1u || it.nopush) ? 1u : 0u};
            else
                S.fail = 1;
        }
    }
}

// one work item of a bin's workgroup (see the banner above).  All RT threads.  LK / LI: 8192-entry LDS arrays.
__device__ __noinline__ void rank_item(const Eng& E, Ctl* c, RankShared& S, uint64_t* LK, uint32_t* LI, uint32_t nf,
                                       uint32_t want, RankItem it) {
    const uint32_t t = threadIdx.x;
    const uint64_t* __restrict__ K = (it.src ? E.ord_key : E.tmp_key) + it.off;
    const uint32_t* __restrict__ I = (it.src ? E.ord_id : E.tmp_id) + it.off;
    uint64_t* __restrict__ K2 = (it.src ? E.tmp_key : E.ord_key) + it.off;
    uint32_t* __restrict__ I2 = (it.src ? E.tmp_id : E.ord_id) + it.off;
    const uint32_t n = it.n, need = it.need;
    if (n <= kDirectMax) {
        // small item: all-pairs out of LDS
        if (t < n) {
            LK[t] = K[t];
            LI[t] = I[t];
        }
        __syncthreads();
        ret_begin(c, S, nf, n - need);
        RetAcc acc{};
        const bool live = t < n;
        uint64_t k = 0;
        uint32_t id = 0, rank = 0;
        if (live) {
            k = LK[t];
            id = LI[t];
            for (uint32_t j = 0; j < n; j++) rank += pair_less(LK[j], LI[j], k, id) ? 1u : 0u;
            if (rank < need) emit_pop(E, c, it.off + rank, k, id);
        }
        ret_put(E, c, S, nf, live && rank >= need, k, id, acc);
        ret_end(c, nf, acc);
        __syncthreads();
        return;
    }
    uint32_t lg = 0;
    while ((8u << lg) <= n && lg < 11) lg++;  // 2^lg <= n / 4, at most kSub sub-bins
    {  // one step finer than n / 4 (shorter ranking loops), tunable for experiments
        const uint32_t lgx = lg + 1u + (uint32_t)g_tune[0];
        lg = lgx < 11u ? lgx : 11u;
    }
    if (n <= kLdsEnt) {
        // ---- the item is read from HBM ONCE (kRegEnt entries per thread, kept in registers), bucketed and ranked in LDS.
        // Inside the item an entry is held as a 64-bit offset — key minus the item's smallest key; or, when all keys of
        // the item are equal (a tie group), id minus its smallest id — so the sub-bin is a shift, not 128-bit
        // arithmetic, and stays monotone in the (key,id) order; pairs inside a sub-bin compare (offset, id).
        // A bin of several thousand entries is shared between it.G workgroups: every one of them reads and counts the
        // whole bin (cheap), then scatters, ranks and emits only its own run of sub-bins.
        uint64_t ek[kRegEnt];
        uint32_t ei[kRegEnt];
        // (index clamped, not predicated: a predicated load compiles to branch + wait per element here — sixteen
        // serialized memory round trips, which is what made this path take 20-45 us)
        prof_begin(E, P_RB_LOAD);
#pragma unroll
        for (int j = 0; j < kRegEnt; j++) {
            const uint32_t i = t + (uint32_t)RT * j, ic = i < n ? i : n - 1;
            ek[j] = K[ic];
            ei[j] = I[ic];
        }
        {
            uint64_t kmn = ~0ull, kmx = 0;
            uint32_t imn = ~0u, imx = 0;
#pragma unroll
            for (int j = 0; j < kRegEnt; j++) {  // (clamped duplicates change nothing)
                const uint32_t im = ei[j] & ID_MASK;
                kmn = ek[j] < kmn ? ek[j] : kmn;
                kmx = ek[j] > kmx ? ek[j] : kmx;
                imn = im < imn ? im : imn;
                imx = im > imx ? im : imx;
            }
            rank_range64(S, kmn, kmx, imn, imx);
        }
        prof_end(E, P_RB_LOAD);
        prof_begin(E, P_RB_COUNT);
        const bool idmode = S.kspan == 0;
        const uint64_t kbase = S.kmin, span = idmode ? (uint64_t)S.ispan : S.kspan;
        const uint32_t ibase = S.imin;
        const uint32_t bits = span ? (uint32_t)(64 - __clzll((long long)span)) : 0u;
        if (lg > bits) lg = bits;  // cannot cut finer than one value
        const uint32_t nsub = 1u << lg, shc = bits - lg;  // (shc <= 63: bits == 64 comes with lg >= 1)
        // a sub-bin too large to rank in place is refined by a work item of its own — unless it cannot be cut further
        // (one id value per sub-bin; in key mode a one-key sub-bin comes back as a tie group, cut by id)
        const bool refinable = (shc > 0 || !idmode) && !it.nopush;
        const uint32_t lo_sub = (uint32_t)((uint64_t)nsub * it.g / it.G), hi_sub = (uint32_t)((uint64_t)nsub * (it.g + 1u) / it.G);
#pragma unroll
        for (int j = 0; j < kRegEnt; j++) {  // offsets replace the keys (registers are short here: no spills wanted)
            ek[j] = idmode ? (uint64_t)((ei[j] & ID_MASK) - ibase) : ek[j] - kbase;
            if (t + (uint32_t)RT * j < n) atomicAdd(&S.cnt[(uint32_t)(ek[j] >> shc)], 1u);
        }
        __syncthreads();
        rank_prefix(S, nsub, need);
        prof_end(E, P_RB_COUNT);
        prof_begin(E, P_RB_SCATTER);
        if (E.prof != nullptr) {  // (diagnostic) largest sub-bin
            uint32_t mx = 0;
            for (uint32_t sb = t; sb < nsub; sb += RT) mx = max(mx, S.off[sb + 1] - S.off[sb]);
            if (mx > 16) atomicMax(&c->dbg_maxsub, mx);
        }
        const uint32_t tsub = S.tsub;
        const bool tsub_pushed = S.off[tsub + 1] - S.off[tsub] > kSubMax && refinable;
        const uint32_t in_hi = hi_sub < tsub + 1u ? hi_sub : tsub + 1u;          // my sub-bins [lo_sub, in_hi) are ranked,
        const uint32_t ret_lo = lo_sub > tsub + 1u ? lo_sub : tsub + 1u;         // [ret_lo, hi_sub) go back to OPEN
        // what THIS workgroup hands back: its sub-bins above the threshold sub-bin, plus that sub-bin's overshoot if it is
        // one of them — unless the sub-bin is refined by a work item of its own (which then reserves for itself)
        ret_begin(c, S, nf, (ret_lo < hi_sub ? S.off[hi_sub] - S.off[ret_lo] : 0u) +
                                ((tsub >= lo_sub && tsub < hi_sub && !tsub_pushed) ? S.off[tsub + 1] - need : 0u));
        const uint32_t n_ord = c->n_ord;
        {
            // Nothing conditional around the LDS reads / returning atomics (a branch per entry means a wait per
            // entry): an entry that does not take part adds zero to an idle word of its own lane.
            uint32_t inm = 0, retm = 0, pos[kRegEnt];
#pragma unroll
            for (int j = 0; j < kRegEnt; j++) {
                const uint32_t sub = (uint32_t)(ek[j] >> shc);
                const bool live = t + (uint32_t)RT * j < n, in = live && sub >= lo_sub && sub < in_hi;
                inm |= (in ? 1u : 0u) << j;
                retm |= (live && sub >= ret_lo && sub < hi_sub ? 1u : 0u) << j;  // the rest of a threshold bin stays in OPEN
                pos[j] = S.off[in ? sub : 0u] + atomicAdd(&S.cnt[in ? sub : (uint32_t)kSub + (t & 63u)], in ? 1u : 0u);
            }
#pragma unroll
            for (int j = 0; j < kRegEnt; j++) {
                if ((inm >> j) & 1u) {
                    LK[pos[j]] = ek[j];
                    LI[pos[j]] = ei[j];
                }
                ek[j] = idmode ? kbase : kbase + ek[j];  // the key back from the offset
            }
            ret_put_many<kRegEnt>(E, S, nf, n_ord, retm, ek, ei);
        }
        __syncthreads();
        prof_end(E, P_RB_SCATTER);
        prof_begin(E, P_RB_ORDER);
        const uint32_t p_lo = S.off[lo_sub], p_hi = in_hi > lo_sub ? S.off[in_hi] : p_lo;
        constexpr int U = 4;  // entries a thread ranks side by side: their LDS reads overlap instead of queueing up
        for (uint32_t p0 = p_lo; p0 < p_hi; p0 += RT * U) {
            uint64_t o[U];
            uint32_t id[U], s0[U], cn[U], rank[U], livem = 0, bigm = 0, trip = 0;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t p = p0 + (uint32_t)u * RT + t;
                livem |= (p < p_hi ? 1u : 0u) << u;
                o[u] = LK[p < p_hi ? p : p_hi - 1];
                id[u] = LI[p < p_hi ? p : p_hi - 1];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t sub = (uint32_t)(o[u] >> shc);
                s0[u] = S.off[sub];
                cn[u] = S.off[sub + 1] - s0[u];
                rank[u] = s0[u];
                if (!((livem >> u) & 1u)) cn[u] = 0;
                if (cn[u] > kSubMax && refinable) {  // refined by the sub-bin's own work item, from the other scratch array
                    bigm |= 1u << u;
                    cn[u] = 0;
                }
                trip = cn[u] > trip ? cn[u] : trip;
            }
            for (uint32_t jj = 0; jj < trip; jj++) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint32_t j = s0[u] + (jj < cn[u] ? jj : 0u);
                    const bool less = pair_less(LK[j], LI[j], o[u], id[u]);
                    rank[u] += (jj < cn[u] && less) ? 1u : 0u;
                }
            }
            uint32_t retm = 0;
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t p = p0 + (uint32_t)u * RT + t;
                o[u] = idmode ? kbase : kbase + o[u];  // the key back from the offset
                if ((bigm >> u) & 1u) {
                    K2[p] = o[u];
                    I2[p] = id[u];
                } else if ((livem >> u) & 1u) {
                    if (rank[u] < need) emit_pop(E, c, it.off + rank[u], o[u], id[u]);
                    else retm |= 1u << u;
                }
            }
            ret_put_many<U>(E, S, nf, n_ord, retm, o, id);
        }
        rank_push(S, it, tsub, need, refinable, lo_sub, in_hi);
        __syncthreads();
        prof_end(E, P_RB_ORDER);
        return;
    }
    // ---- larger than LDS: three streaming passes over the slice, 8 loads in flight per thread
    {
        u128 vmin = ~(u128)0, vmax = 0;
        for (uint32_t b0 = 0; b0 < n; b0 += kLdsEnt) {
            uint64_t ek[kRegEnt];
            uint32_t ei[kRegEnt];
#pragma unroll
            for (int j = 0; j < kRegEnt; j++) {
                const uint32_t i = b0 + t + (uint32_t)RT * j, ic = i < n ? i : n - 1;
                ek[j] = K[ic];
                ei[j] = I[ic];
            }
#pragma unroll
            for (int j = 0; j < kRegEnt; j++)
                if (b0 + t + (uint32_t)RT * j < n) {
                    const u128 v = comp_of(ek[j], ei[j]);
                    vmin = v < vmin ? v : vmin;
                    vmax = v > vmax ? v : vmax;
                }
        }
        rank_range(S, vmin, vmax);
    }
    const u128 base = ((u128)S.vmin_hi << 64) | S.vmin_lo;
    const uint32_t bits = S.bits;
    if (lg > bits) lg = bits;
    const uint32_t nsub = 1u << lg, shc = bits - lg;
    for (uint32_t b0 = 0; b0 < n; b0 += kLdsEnt) {
        uint64_t ek[kRegEnt];
        uint32_t ei[kRegEnt];
#pragma unroll
        for (int j = 0; j < kRegEnt; j++) {
            const uint32_t i = b0 + t + (uint32_t)RT * j, ic = i < n ? i : n - 1;
            ek[j] = K[ic];
            ei[j] = I[ic];
        }
#pragma unroll
        for (int j = 0; j < kRegEnt; j++)
            if (b0 + t + (uint32_t)RT * j < n) atomicAdd(&S.cnt[sub_of(ek[j], ei[j], base, shc, nsub)], 1u);
    }
    __syncthreads();
    rank_prefix(S, nsub, need);
    const uint32_t tsub = S.tsub;
    const bool tsub_pushed = S.off[tsub + 1] - S.off[tsub] > kSubMax && shc > 0;
    ret_begin(c, S, nf, (n - S.off[tsub + 1]) + (tsub_pushed ? 0u : S.off[tsub + 1] - need));
    RetAcc acc{};
    const uint32_t n_ord_all = c->n_ord;
    for (uint32_t b0 = 0; b0 < n; b0 += kLdsEnt) {
        uint64_t ek[kRegEnt];
        uint32_t ei[kRegEnt];
#pragma unroll
        for (int j = 0; j < kRegEnt; j++) {
            const uint32_t i = b0 + t + (uint32_t)RT * j, ic = i < n ? i : n - 1;
            ek[j] = K[ic];
            ei[j] = I[ic];
        }
        uint32_t retm = 0;
#pragma unroll
        for (int j = 0; j < kRegEnt; j++) {
            const bool live = b0 + t + (uint32_t)RT * j < n;
            uint32_t sub = 0;
            if (live) {
                sub = sub_of(ek[j], ei[j], base, shc, nsub);
                if (sub <= tsub) {
                    const uint32_t p = S.off[sub] + atomicAdd(&S.cnt[sub], 1u);
                    K2[p] = ek[j];
                    I2[p] = ei[j];
                }
            }
            retm |= (live && sub > tsub ? 1u : 0u) << j;
        }
        // (one batched hand-back per chunk: a ret_put per entry waits for its own slot-number load — sixteen memory round
        // trips per chunk, hundreds of chunks when the bin is a tie group of millions)
        ret_put_many<kRegEnt>(E, S, nf, n_ord_all, retm, ek, ei);
    }
    __syncthreads();
    const uint32_t m = S.off[tsub + 1];
    for (uint32_t p0 = 0; p0 < m; p0 += RT) {
        const uint32_t p = p0 + t;
        bool live = p < m;
        uint64_t k = 0;
        uint32_t id = 0, rank = 0;
        if (live) {
            k = K2[p];
            id = I2[p];
            const uint32_t sub = sub_of(k, id, base, shc, nsub);
            const uint32_t s0 = S.off[sub], e0 = S.off[sub + 1];
            if (e0 - s0 > kSubMax && shc > 0) {
                live = false;  // ranked by the sub-bin's own work item
            } else {
                rank = s0;
                for (uint32_t j = s0; j < e0; j++) rank += pair_less(K2[j], I2[j], k, id) ? 1u : 0u;
                if (rank < need) emit_pop(E, c, it.off + rank, k, id);
            }
        }
        ret_put(E, c, S, nf, live && rank >= need, k, id, acc);
    }
    rank_push(S, it, tsub, need, shc > 0);
    ret_end(c, nf, acc);
    __syncthreads();
}

__global__ __launch_bounds__(RT) void k_rank(const Eng* __restrict__ engs) {
    const Eng& E = engs[blockIdx.y];
    Ctl* c = E.ctl;
    if (c->done) return;
    Stamp stamp(E, P_RANK);
    __shared__ RankShared S;
    extern __shared__ __attribute__((aligned(16))) uint8_t rank_lds[];  // 8192 keys, then 8192 ids (96 KB)
    uint64_t* LK = reinterpret_cast<uint64_t*>(rank_lds);
    uint32_t* LI = reinterpret_cast<uint32_t*>(rank_lds + (size_t)kLdsEnt * 8);
    const uint32_t t = threadIdx.x;
    const uint32_t nf = c->cur_f;  // what the batch does not take goes back where it came from
    const uint32_t tseg = c->tseg, want = c->want;  // (tseg: the segment holding the batch's last entry — the threshold bin)
    const uint32_t n_big = c->n_big, n_ord = c->n_ord;
    if (blockIdx.x == gridDim.x - 1) {
        // housekeeping for the launches that read these arrays while every workgroup is looking (k_sel_collect): FRONT's
        // histogram minus what this pop takes — the bins below the threshold bin leave entirely, the threshold bin keeps
        // its overshoot (k_commit adds the children later in this iteration) — and the scratch counters back to zero
        const uint32_t bstar = c->bstar, pre_b = c->pre_b, cn_star = c->cn_star;
        if (want != 0) {
            for (uint32_t bin = t; bin < bstar; bin += RT) E.hist[bin] = 0;
            if (t == 0) E.hist[bstar] = cn_star - (want - pre_b);
        }
        for (uint32_t i = t; i < (uint32_t)kSegs; i += RT) E.fill[i] = 0;
        if (c->giant)
            for (uint32_t i = t; i < (uint32_t)(kMaxLevels * kSub); i += RT) E.subhist[i] = 0;
    }
    // ---- entries of small bins (most bins, about half the entries): one thread each, the whole grid at once
    {
        Stamp sub(E, P_RANK_SMALL);  // (profile only: the two halves of this launch get their own slots)
        for (uint32_t p0 = blockIdx.x * RT; p0 < n_ord; p0 += gridDim.x * RT)
            rank_small_entries(E, c, S, LK, LI, nf, p0, n_ord, want);
    }
    Stamp sub2(E, P_RANK_BIG);
    // ---- bins of more than kTinyBin entries: one workgroup each.  The first n_ord / RT workgroups are busy with the pass
    // above, so the large bins start at workgroup 64
    for (uint32_t bi = (blockIdx.x + gridDim.x - 64u) % gridDim.x; bi < n_big; bi += gridDim.x) {
        const uint32_t unit = E.big_list[bi];  // segment | share << 16 | shares << 20
        const uint32_t f = unit & 0xFFFFu;
        const uint32_t o = E.pre[f], n = E.pre[f + 1] - o;
        __syncthreads();
        if (t == 0) {
            // entries of this bin that belong to the batch: all of it below the threshold bin
            S.stack[0] = RankItem{o, n, (f == tseg) ? want - o : n, 0u, (unit >> 16) & 15u, unit >> 20, 0u};
            S.sp = 1;
            S.fail = 0;
        }
        for (;;) {
            __syncthreads();
            const uint32_t sp = S.sp;
            if (sp == 0 || sp > (uint32_t)kRankStack) break;
            const RankItem it = S.stack[sp - 1];
            __syncthreads();
            if (t == 0) S.sp = sp - 1;
            __syncthreads();
            rank_item(E, c, S, LK, LI, nf, want, it);
        }
        if (t == 0 && (S.fail || S.sp != 0)) c->failed = 1;  // (cannot happen: every refinement level narrows the range)
    }
}

// ---------------------------------------------------------------------------------------------
// expand: gather the popped parents by id, write children rows + node fields + hash/solved/heuristic
// ---------------------------------------------------------------------------------------------
// The pop's single-thread epilogue (thread 0 of the expansion's workgroup 0): goal bookkeeping and the state the rest
// of the iteration — and the next one — starts from.
__device__ __forceinline__ void close_pop(const Eng& E, Ctl* c, const IterState& S0, uint32_t it, uint32_t want,
                                          uint32_t npop, uint32_t base, bool fail) {
    IterState N = S0;
    if (E.sem == DCA_SEM_PY) {
        // astar.py:73,421: any solved node among the popped ends the search after this iteration;
        // answer = solved popped node of smallest path cost, first in pop order on ties (327-333)
        const unsigned long long gb = c->goal_best;
        if (gb != ~0ull) {
            c->stop_after = 1;
            c->goal_id = E.pop_id[(uint32_t)(gb & 0xFFFFFFFFull)] & ID_MASK;
        }
    } else {
        // cpp:185-208
        const uint32_t fs = c->first_solved;
        if (fs != NIL) {
            const float cst = (float)cost_of_key(E.pop_key[fs]);
            if (E.B == 1) {
                N.best_id = E.pop_id[fs] & ID_MASK;
                N.best_cost = cst;
                N.has_best = 1;
                c->stop_after = 1;
            } else if (!S0.has_best || S0.best_cost > cst) {
                N.best_id = E.pop_id[fs] & ID_MASK;
                N.best_cost = cst;
                N.has_best = 1;
            }
        }
        if (S0.has_best && (float)cost_of_key(E.pop_key[0]) >= N.best_cost) c->stop_after = 1;
    }
    if (fail) {  // node pool exhausted
        c->failed = 1;
        c->done = 1;
        c->front_dead.v -= want;  // the pop is void: OPEN's reported size stays that of the last complete iteration
        return;
    }
    const uint32_t m = npop * (uint32_t)E.A;
    N.npop = npop;
    N.m = m;
    N.base = base;
    N.pool_n = base + m;
    c->S[(it + 1) & 1] = N;
    c->gen += (int64_t)m;
    c->expanded += npop;
}

constexpr uint8_t F_KEEP = 1, F_MIN = 2, F_NEW = 4;

// Word `w` (bytes 4w .. 4w+3) of the child that move `a` makes of a parent whose row is spread over the 16 lanes of this
// lane's group, one 4-byte word per lane (`pw`; bytes past the row are 0).  On-chip only: the gather runs over lane
// shuffles.  All 64 lanes must call it together (shuffles read from lanes of the caller's own 16-lane group).
template <int ENV, int DIM>
__device__ __forceinline__ uint32_t child_word_from_parent(uint32_t pw, uint32_t w, uint32_t a, const uint8_t* ltab) {
    using EV = EnvT<ENV, DIM>;
    const int gbase = (int)(threadIdx.x & 63u) & ~15;
    uint32_t out = 0;
    if constexpr (ENV == DCA_ENV_CUBE3) {
#pragma unroll
        for (uint32_t b = 0; b < 4; b++) {
            const uint32_t i = 4 * w + b;
            const uint32_t p = i < (uint32_t)EV::D ? ltab[a * EV::D + i] : 0u;
            const uint32_t v = (uint32_t)__shfl((int)pw, gbase + (int)(p >> 2));
            if (i < (uint32_t)EV::D) out |= ((v >> (8 * (p & 3))) & 0xFFu) << (8 * b);
        }
    } else if constexpr (ENV == DCA_ENV_LIGHTSOUT) {
#pragma unroll
        for (uint32_t b = 0; b < 4; b++) {
            const uint32_t i = 4 * w + b;
            if (i < (uint32_t)EV::D) {
                const uint32_t v = (pw >> (8 * b)) & 0xFFu;
                out |= (lightsout_flip(DIM, (int)a, (int)i) ? ((v + 1u) & 1u) : v) << (8 * b);
            }
        }
    } else {
        // the blank: first zero byte of the row (n_puzzle.py:51-53), found group-wide
        uint32_t z = 0xFFFFu;
#pragma unroll
        for (int b = 3; b >= 0; b--)
            if (4 * w + b < (uint32_t)EV::D && ((pw >> (8 * b)) & 0xFFu) == 0u) z = 4 * w + b;
        for (int o = 8; o > 0; o >>= 1) {
            const uint32_t y = (uint32_t)__shfl_xor((int)z, o);
            z = y < z ? y : z;
        }
        z = z < (uint32_t)EV::D ? z : 0u;
        const uint32_t sw = (uint32_t)npuzzle_swap(DIM, (int)z, (int)a);
#pragma unroll
        for (uint32_t b = 0; b < 4; b++) {
            const uint32_t i = 4 * w + b;
            const uint32_t src = i < (uint32_t)EV::D ? ((i == z) ? sw : i) : 0u;  // next[z] = cur[s]; next[s] = 0
            const uint32_t v = (uint32_t)__shfl((int)pw, gbase + (int)(src >> 2));
            if (i < (uint32_t)EV::D) out |= ((i == sw) ? 0u : ((v >> (8 * (src & 3))) & 0xFFu)) << (8 * b);
        }
    }
    return out;
}

constexpr int kEngTile = 16;  // parents per workgroup: a 20 000-parent batch then fills the chip (1250 workgroups)
// PROBE: the CLOSED probe of the batch's children (what k_probe does in a launch of its own) runs here, from the rows the
// tile holds in LDS — one launch, one boundary and one 13 MB re-read of the child rows less per iteration.  The one thing
// k_probe could rely on and this kernel cannot: a representative inserted by ANOTHER workgroup of the same launch has no
// row in HBM yet.  It needs none: its id says which popped parent and which move made it (id = base + rank * A + move),
// and the parent's row was written an iteration ago — the tile rebuilds the representative from it (16 lanes per pending
// comparison) and compares exactly.
template <int ENV, int DIM, int OH, bool PROBE>
__global__ __launch_bounds__(kThreads) void k_expand(const Eng* __restrict__ engs, int heur_id, int write_nn) {
    const Eng& E = engs[blockIdx.y];
    using EV = EnvT<ENV, DIM>;
    using TL = Tile<ENV, DIM, kEngTile>;
    constexpr int kTileParents = kEngTile;  // shadows the stand-alone kernels' 64 (32 / 64 parents per workgroup measured SLOWER
                                            // also for the launches that write one-hot rows: profiles/r06_engine_ab.txt)
    constexpr int TP = kEngTile;
    Ctl* c = E.ctl;
    if (c->done) return;
    Stamp stamp(E, P_EXPAND);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* lpar = smem;
    uint8_t* ltab = smem + TL::PAR_BYTES;
    uint8_t* lst = smem + TL::LDS_BYTES;  // child rows of the tile, laid out exactly like their HBM destination
    __shared__ uint32_t l_pid[TP], l_g[TP];
    __shared__ uint32_t l_qn, l_qcc[PROBE ? kThreads : 1], l_qrep[PROBE ? kThreads : 1], l_qres[PROBE ? kThreads : 1];
    __shared__ uint32_t l_ohq;  // one-hot rows: next 1 KiB piece (one wave-wide 16-byte store) nobody has claimed yet
    static_assert((EV::D + 3) / 4 < 16, "no spare lane per row for the parent's path cost");
    // batch geometry: every workgroup derives it from the state the previous iteration left (S[iters & 1]) and the
    // pop that k_rank just finished; workgroup 0 also records it for the rest of the iteration (close_pop)
    const uint32_t it = (uint32_t)c->iters;
    const IterState S0 = c->S[it & 1];
    const uint32_t want = c->want;
    uint32_t npop = want;
    if (E.sem == DCA_SEM_CPP) {
        const uint32_t fs = c->first_solved;
        if (fs != NIL) npop = fs + 1;  // cpp:203 break at the first solved node popped
    }
    const uint32_t base = (S0.pool_n + 15u) & ~15u;  // 16-aligned ids => 16-byte aligned child rows for every D
    const bool fail = (uint64_t)base + (uint64_t)npop * (uint64_t)EV::A > (uint64_t)E.max_nodes;
    const uint32_t cur_new = c->cur_f;  // FRONT is edited in place: put-backs and children are appended to it
    if (blockIdx.x == 0 && threadIdx.x == 0) close_pop(E, c, S0, it, want, npop, base, fail);
    if (fail) return;
    const uint32_t r0 = blockIdx.x * kTileParents;
    if (E.sem == DCA_SEM_CPP && r0 + kTileParents > npop) {
        // cpp:203 `break`: entries selected after the first solved node were never popped — put them back
        uint32_t r = r0 + threadIdx.x;
        bool back = threadIdx.x < kTileParents && r >= npop && r < want;
        open_append(E, c, cur_new, back, back ? E.pop_key[r] : 0, back ? E.pop_id[r] : 0);
        if (back) atomicAdd(&E.hist[bin_of(E.pop_key[r], c->sel_kmin, c->shift)], 1u);  // FRONT's histogram is incremental
    }
    if (r0 >= npop) return;
    const uint32_t np = min((uint32_t)kTileParents, npop - r0);
    if (OH != 0 && threadIdx.x == 0) l_ohq = 0;
    {
        // gather the popped rows by node id.  One lane per (parent, 4-byte word): 16 lanes cover a row, so
        // a 256-thread block fetches 16 rows per round and the 4 rounds are issued back to back.
        constexpr int WPR = (EV::D + 3) / 4;  // words per row (<= 14)
        static_assert(WPR <= 16, "row wider than 64 bytes");
        const uint32_t w = threadIdx.x & 15;  // 16 lanes per row, 16 rows per round
        // (a spare lane of every row fetches the parent's path cost in the same round trip as the row itself: the
        // per-child loop below used to wait for it — a dependent random access — before it could write anything)
#pragma unroll
        for (uint32_t r = threadIdx.x >> 4; r < (uint32_t)TP; r += 16) {
        if (r < np && w == WPR) {
            const uint32_t pid = E.pop_id[r0 + r] & ID_MASK;
            l_pid[r] = pid;
            l_g[r] = (uint32_t)E.g[pid];
        }
        if (r < np && w < WPR) {
            const uint8_t* row = E.state + (size_t)(E.pop_id[r0 + r] & ID_MASK) * EV::D;
            uint32_t v = 0;
            // rows start at id*D: read byte-wise only where a word would cross the row end
            if (4 * w + 4 <= EV::D) {
                __builtin_memcpy(&v, row + 4 * w, 4);
            } else {
                for (int b = 0; 4 * w + b < EV::D; b++) v |= (uint32_t)row[4 * w + b] << (8 * b);
            }
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (4 * w + b < EV::D) lpar[r * EV::D + 4 * w + b] = (uint8_t)(v >> (8 * b));
        }
        }
    }
    if constexpr (ENV == DCA_ENV_CUBE3) stage_tables<ENV, DIM>(ltab, lpar, np);
    __syncthreads();
    if constexpr (ENV != DCA_ENV_CUBE3) {
        stage_tables<ENV, DIM>(ltab, lpar, np);
        __syncthreads();
    }
    TL t{lpar, ltab};
    const uint32_t nchild = np * EV::A;
    const uint32_t j0 = r0 * EV::A;  // first child index of the tile within the batch

    // One-hot rows of the tile's children (pytorch_models.py:49-52), from the child rows staged in `lst`.  The rows of a tile are
    // one contiguous run of the batch's one-hot buffer; it is cut into 1 KiB pieces — one wave-wide 16-byte store each — which
    // the waves CLAIM from an LDS counter.  Every store comes after the probe: all workgroups of the launch are resident at
    // once, they probe at the same time and then store at the same time, and the launch takes its stores (57-63 us for fp32
    // rows at B = 20 000) PLUS its probe (~28 us).  Round 6 tried to overlap the two — the wave of the per-child round that holds
    // no child storing under the others' probe, unpaced (110-141 us instead of 97: the probe's loads queue behind the stores in
    // the CU's one memory pipe) and paced (level) — and larger tiles (32 / 64 parents: 108 / 145 us); profiles/r06_engine_ab.txt.
    auto oh_emit = [&]() {
        if constexpr (OH != 0) {
            constexpr uint32_t ROW = EV::D * EV::DEPTH;
            constexpr uint32_t EPC = 16 / OH;
            const uint32_t te = nchild * ROW;
            const uint32_t one16 = E.oh_dtype == DCA_DT_F16 ? 0x3C00u : 0x3F80u;
            uint8_t* goh = E.onehot + (size_t)j0 * ROW * OH;
            const uint32_t nch = (te + EPC - 1) / EPC;
            // (one read of the staged row per position; a lane that runs one row past the tile reads the slack behind it — its
            // chunk is never stored)
            auto staged_nnet = [&](uint32_t i) -> uint32_t {
                const uint32_t b = lst[i];
                return ENV == DCA_ENV_CUBE3 ? (b * 57u) >> 9 : b;
            };
            for (;;) {
                uint32_t piece = 0;
                if ((threadIdx.x & 63u) == 0) piece = atomicAdd(&l_ohq, 1u);
                piece = (uint32_t)__builtin_amdgcn_readfirstlane((int)piece);
                if (piece * 64u >= nch) break;
                const uint32_t q = piece * 64u + (threadIdx.x & 63u);
                if (q >= nch) continue;
                uint32_t e0 = q * EPC;
                if constexpr (ENV == DCA_ENV_CUBE3 && OH == 4) {
                    if (e0 + EPC <= te) {  // one or two stickers and four compares per chunk (dca_tile.h: cube3_onehot32_chunk)
                        const uint32_t P0 = (q << 1) / 3u, phase = (q << 1) - 3u * P0;
                        uint32_t w[4];
                        cube3_onehot32_chunk(phase, staged_nnet(P0), phase == 2u ? staged_nnet(P0 + 1u) : 0u, w);
                        store16(goh + (size_t)e0 * OH, w, true);
                        continue;
                    }
                }
                if constexpr (ENV == DCA_ENV_CUBE3 && OH == 2) {
                    if (e0 + EPC <= te) {  // two stickers and a lookup per chunk (dca_tile.h: cube3_onehot16_chunk)
                        const uint32_t P0 = (q << 2) / 3u, phase = (q << 2) - 3u * P0;
                        uint32_t w[4];
                        cube3_onehot16_chunk(phase, staged_nnet(P0), staged_nnet(P0 + 1u), one16, w);
                        store16(goh + (size_t)e0 * OH, w, true);
                        continue;
                    }
                }
                uint32_t cch = e0 / ROW, e = e0 - cch * ROW;
                uint32_t pos = e / EV::DEPTH, col = e - pos * EV::DEPTH;
                uint32_t r = cch / EV::A, a = cch - r * EV::A;
                uint32_t nb = staged_nnet(cch * EV::D + pos);
                uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
                for (uint32_t k = 0; k < EPC; k++) {
                    bool hot = (nb == col) && (e0 + k < te);
                    if constexpr (OH == 4)
                        w[k] = hot ? 0x3F800000u : 0u;
                    else
                        w[k >> 1] |= (hot ? one16 : 0u) << (16 * (k & 1));
                    if (++col == EV::DEPTH) {
                        col = 0;
                        if (++pos == EV::D) {
                            pos = 0;
                            if (++a == EV::A) {
                                a = 0;
                                ++r;
                            }
                        }
                        nb = staged_nnet((r * EV::A + a) * EV::D + pos);
                    }
                }
                uint8_t* dst = goh + (size_t)e0 * OH;
                if (e0 + EPC <= te) {
                    store16(dst, w, true);
                } else {
                    for (uint32_t k = 0; e0 + k < te; k++) {
                        if constexpr (OH == 4)
                            reinterpret_cast<uint32_t*>(dst)[k] = w[k];
                        else
                            reinterpret_cast<uint16_t*>(dst)[k] = (uint16_t)(w[k >> 1] >> (16 * (k & 1)));
                    }
                }
            }
        }
    };

    // per child: hash, is_solved, node fields, built-in heuristic (rounds of 256 children; uniform: the probe below
    // synchronises the workgroup inside a round)
    if (PROBE && threadIdx.x == 0) l_qn = 0;
    for (uint32_t cc0 = 0; cc0 < nchild; cc0 += kThreads) {
        const uint32_t cc = cc0 + threadIdx.x;
        const bool cvalid = cc < nchild;
        uint64_t h = 0;
        uint64_t roww[(EV::D + 7) / 8];  // the child's row, 8 bytes a word (PROBE: compared against representatives' rows)
        if (cvalid) {
        uint32_t r = cc / EV::A, a = cc - r * EV::A;
        uint64_t sum = 0;
        h = hash_init(EV::D);
        uint32_t manh = 0;
        bool ok = true;
#pragma unroll
        for (int k = 0; k < EV::D; k += 8) {
            uint64_t w = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (k + j < EV::D) {
                    uint32_t b = t.child_byte(r, a, k + j);
                    uint32_t goal = goal_byte(ENV, EV::D, k + j);
                    ok &= (b == goal);
                    w |= (uint64_t)b << (8 * j);
                    sum += (uint64_t)b * (uint64_t)(7 * (k + j) + 3);
                    if constexpr (ENV == DCA_ENV_NPUZZLE) manh += manhattan_term(DIM, (uint32_t)(k + j), b);
                }
            }
            h = hash_word(h, w);
            roww[k >> 3] = w;
            // stage the gathered bytes (rows start on even offsets when D is even: 2-byte stores; else bytes)
            uint8_t* dst = lst + cc * EV::D + k;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                if (k + j + 1 < EV::D && (EV::D % 2) == 0) {
                    *reinterpret_cast<uint16_t*>(dst + j) = (uint16_t)(w >> (8 * j));
                } else {
                    if (k + j < EV::D) dst[j] = (uint8_t)(w >> (8 * j));
                    if (k + j + 1 < EV::D) dst[j + 1] = (uint8_t)(w >> (8 * j + 8));
                }
            }
        }
        h = hash_final(h);
        const uint32_t j = j0 + cc, id = base + j, pid = l_pid[r];
        const uint32_t gp = l_g[r];
        if (a == 0) E.pop_g[r0 + r] = gp;  // the parents' path costs in pop order: what the dedup / push kernels read
        if constexpr (!PROBE) {
            E.child_hash[j] = h;
            E.child_multi[j] = 0;  // (PROBE: the marks are cleared by k_commit after it has read them — another workgroup of this
                                   // launch may set this child's mark before or after this point)
        }
        E.g[id] = (int32_t)(gp + 1u);  // path cost + unit transition cost (astar.py:125-126 / cpp:219)
        E.parent[id] = pid;
        E.move[id] = (uint8_t)a;
        E.solved[id] = ok ? 1 : 0;
        if (heur_id >= 0) E.child_h[j] = heur_from(heur_id, sum, h, manh);
        }
        if constexpr (PROBE) {
            // ---- find-or-insert the CLOSED slot of this round's children (k_probe's loop, rows taken from LDS)
            constexpr int NWD = (EV::D + 3) / 4;
            const uint32_t j = j0 + cc, id = base + j;
            const uint64_t tag = h >> 32;
            bool active = cvalid, inserted = false, paused = false;
            uint32_t slot = (uint32_t)h & E.tab_mask, v0 = GINF, rep_id = 0, probes = 0, qi = 0;
            __syncthreads();  // the round's rows are staged (lst) — and l_qn is zero
            for (;;) {
                if (active && !paused) {
                    for (;;) {
                        // look first, claim second; one 16-byte load fetches the entry with its value (see k_probe)
                        const uint4 sl = *reinterpret_cast<const uint4*>(&E.tab[slot]);
                        uint64_t e = ((uint64_t)sl.y << 32) | sl.x;
                        v0 = sl.z;
                        if (e == EMPTY) {
                            const uint64_t old = atomicCAS((unsigned long long*)&E.tab[slot].entry, (unsigned long long)EMPTY,
                                                           (unsigned long long)((tag << 32) | id));
                            if (old == EMPTY) {
                                inserted = true;
                                active = false;
                                break;
                            }
                            e = old;  // claimed meanwhile by another child of this batch: its value is still GINF, as loaded
                        }
                        if ((e >> 32) == tag) {
                            const uint32_t rep = (uint32_t)e;
                            if (rep >= base) {
                                // a representative of THIS batch: no row in HBM yet — the workgroup rebuilds it from its parent
                                qi = atomicAdd(&l_qn, 1u);
                                l_qcc[qi] = cc;
                                l_qrep[qi] = rep;
                                paused = true;
                                break;
                            }
                            // exact key equality against the representative's state bytes (State.__eq__, cube3.py:23-24)
                            const uint8_t* rrow = E.state + (size_t)rep * EV::D;
                            uint32_t diff = 0;
#pragma unroll
                            for (int k = 0; k < NWD; k++) {
                                uint32_t v = 0;
                                if (4 * k + 4 <= EV::D) {
                                    __builtin_memcpy(&v, rrow + 4 * k, 4);
                                } else {
                                    for (int b = 0; 4 * k + b < EV::D; b++) v |= (uint32_t)rrow[4 * k + b] << (8 * b);
                                }
                                diff |= v ^ (uint32_t)(roww[k >> 1] >> (32 * (k & 1)));  // (bytes past the row are 0 in both)
                            }
                            if (diff == 0) {
                                rep_id = rep;
                                active = false;
                                break;
                            }
                        }
                        slot = (slot + 1) & E.tab_mask;
                        if (++probes > E.tab_mask) {  // table full (cannot happen while pool <= cap/2)
                            c->failed = 1;
                            active = false;
                            break;
                        }
                    }
                }
                __syncthreads();
                const uint32_t nq = l_qn;
                if (nq == 0) break;
                // pending comparisons, 16 at a time: lane group g rebuilds the representative of item q0 + g from its parent
                for (uint32_t q0 = 0; q0 < nq; q0 += kThreads / 16) {
                    const uint32_t item = q0 + (threadIdx.x >> 4), w = threadIdx.x & 15u;
                    const bool iv = item < nq;
                    const uint32_t itc = iv ? item : 0u;
                    const uint32_t rrel = l_qrep[itc] - base;
                    const uint32_t rk = rrel / (uint32_t)EV::A, a2 = rrel - rk * (uint32_t)EV::A;
                    const uint32_t ppid = E.pop_id[rk] & ID_MASK;
                    const uint8_t* prow = E.state + (size_t)ppid * EV::D;
                    uint32_t pw = 0;
                    if (w < (uint32_t)NWD) {
                        if (4 * w + 4 <= (uint32_t)EV::D) {
                            __builtin_memcpy(&pw, prow + 4 * w, 4);
                        } else {
                            for (uint32_t b = 0; 4 * w + b < (uint32_t)EV::D; b++) pw |= (uint32_t)prow[4 * w + b] << (8 * b);
                        }
                    }
                    const uint32_t cw = child_word_from_parent<ENV, DIM>(pw, w, a2, ltab);
                    const uint8_t* mrow = lst + l_qcc[itc] * EV::D;
                    uint32_t mv = 0;
                    for (uint32_t b = 0; b < 4 && 4 * w + b < (uint32_t)EV::D; b++) mv |= (uint32_t)mrow[4 * w + b] << (8 * b);
                    uint32_t diff = w < (uint32_t)NWD ? (cw ^ mv) : 0u;
                    for (int o = 8; o > 0; o >>= 1) diff |= (uint32_t)__shfl_xor((int)diff, o);
                    if (iv && w == 0) l_qres[item] = diff == 0 ? 1u : 0u;
                }
                __syncthreads();
                if (paused) {
                    paused = false;
                    if (l_qres[qi]) {
                        rep_id = l_qrep[qi];
                        active = false;
                    } else {  // same tag, another state: keep probing
                        slot = (slot + 1) & E.tab_mask;
                        probes++;
                    }
                }
                __syncthreads();
                if (threadIdx.x == 0) l_qn = 0;
                __syncthreads();
            }
            // chain hook (see k_probe): the child that claimed an empty slot needs none
            if (cvalid) {
                uint32_t prev = NIL;
                if (!inserted) {
                    const uint32_t old_head = atomicExch(&E.tab[slot].head, id);
                    if (old_head >= base)
                        prev = old_head - base;
                    else if (rep_id >= base)
                        prev = rep_id - base;
                }
                E.child_next[j] = prev;
                E.child_slot[j] = slot;
                E.child_v0[j] = v0;
                E.child_flags[j] = inserted ? F_NEW : 0;
                if (prev != NIL) {
                    E.child_multi[j] = 1;
                    E.child_multi[prev] = 1;
                }
            }
        }
    }

    // child rows -> node pool (final place), network-input rows -> batch buffer: straight 16-byte copies of the
    // staged tile (the gathers were paid once, in the per-child loop above).  The network-input rows are written only when
    // somebody reads them (write_nn): not for a built-in heuristic (evaluated above, from the tile) and not in dedup-first
    // stepping (k_pack writes the kept children's rows from the pool) — 13 MB per batch-20 000 cube3 launch otherwise.
    __syncthreads();
    if (!write_nn) {
        const uint32_t tb = nchild * EV::D;
        uint8_t* gpool = E.state + ((size_t)base + j0) * EV::D;
        const uint32_t nfull = tb >> 4;
        for (uint32_t q = threadIdx.x; q < nfull; q += kThreads)
            reinterpret_cast<uint4*>(gpool)[q] = reinterpret_cast<const uint4*>(lst)[q];
        for (uint32_t bq = (nfull << 4) + threadIdx.x; bq < tb; bq += kThreads) gpool[bq] = lst[bq];
    } else {
        const uint32_t tb = nchild * EV::D;
        uint8_t* gpool = E.state + ((size_t)base + j0) * EV::D;
        uint8_t* gnn = E.nnet_in + (size_t)j0 * EV::D;
        const uint32_t nfull = tb >> 4;
        for (uint32_t q = threadIdx.x; q < nfull; q += kThreads) {
            const uint4 v = reinterpret_cast<const uint4*>(lst)[q];
            reinterpret_cast<uint4*>(gpool)[q] = v;
            if constexpr (ENV == DCA_ENV_CUBE3) {
                // sticker // 9 on 16 bytes at once: (b*57)>>9 per byte, two bytes per 32-bit multiply
                auto div9 = [](uint32_t x) {
                    // quotients are < 8: keep 3 bits per field (bits above come from the neighbouring 16-bit lane)
                    uint32_t lo = (((x & 0x00FF00FFu) * 57u) >> 9) & 0x00070007u;
                    uint32_t hi = ((((x >> 8) & 0x00FF00FFu) * 57u) >> 9) & 0x00070007u;
                    return lo | (hi << 8);
                };
                reinterpret_cast<uint4*>(gnn)[q] = make_uint4(div9(v.x), div9(v.y), div9(v.z), div9(v.w));
            } else {
                reinterpret_cast<uint4*>(gnn)[q] = v;
            }
        }
        for (uint32_t bq = (nfull << 4) + threadIdx.x; bq < tb; bq += kThreads) {
            const uint32_t b = lst[bq];
            gpool[bq] = (uint8_t)b;
            gnn[bq] = (uint8_t)(ENV == DCA_ENV_CUBE3 ? (b * 57u) >> 9 : b);
        }
    }

    oh_emit();
}

// ---------------------------------------------------------------------------------------------
// dedup A: find-or-insert the CLOSED slot of every child, chain the child to it
// ---------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_probe(const Eng* __restrict__ engs) {
    const Eng& E = engs[blockIdx.y];
    Ctl* c = E.ctl;
    if (c->done) return;
    Stamp stamp(E, P_PROBE);
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    const IterState& S1 = st_next(c);
    const uint32_t m = S1.m, base = S1.base;
    if (j >= m) return;
    constexpr int NW = (D + 3) / 4;
    const uint32_t id = base + j;
    const uint64_t h = E.child_hash[j];
    const uint64_t tag = h >> 32;
    // this child's row, as words (the row was written by k_expand; the tail word is masked)
    uint32_t mine[NW];
    {
        const uint8_t* row = E.state + (size_t)id * D;
#pragma unroll
        for (int k = 0; k < NW; k++) {
            if (4 * k + 4 <= D) {
                __builtin_memcpy(&mine[k], row + 4 * k, 4);
            } else {
                mine[k] = 0;
                for (int b = 0; 4 * k + b < D; b++) mine[k] |= (uint32_t)row[4 * k + b] << (8 * b);
            }
        }
    }
    bool inserted = false;
    uint32_t slot = (uint32_t)h & E.tab_mask;
    uint32_t v0 = GINF, rep_id = 0;
    for (uint32_t probes = 0;; probes++) {
        // look first, claim second (a compare-and-swap on every probed slot — one round trip instead of two for a new
        // state — was measured and lost badly: 61 us vs 26 us per launch; returning atomics are that much dearer here).
        // ONE 16-byte load fetches the slot's entry together with its value: entries never change once written and
        // values only change in k_commit, so a line cached by an earlier reader on this CU is either current or shows an
        // EMPTY entry that the compare-and-swap then corrects.
        const uint4 sl = *reinterpret_cast<const uint4*>(&E.tab[slot]);
        uint64_t e = ((uint64_t)sl.y << 32) | sl.x;
        v0 = sl.z;  // the slot's value before this batch (GINF while nothing was recorded)
        if (e == EMPTY) {
            const uint64_t old = atomicCAS((unsigned long long*)&E.tab[slot].entry, (unsigned long long)EMPTY,
                                           (unsigned long long)((tag << 32) | id));
            if (old == EMPTY) {
                inserted = true;
                break;
            }
            e = old;  // claimed meanwhile by another child of this batch: its value is still GINF, as loaded
        }
        if ((e >> 32) == tag) {
            // exact key equality against the representative's state bytes (State.__eq__, cube3.py:23-24)
            const uint8_t* rep = E.state + (size_t)(uint32_t)e * D;
            uint32_t diff = 0;
#pragma unroll
            for (int k = 0; k < NW; k++) {
                uint32_t v;
                if (4 * k + 4 <= D) {
                    __builtin_memcpy(&v, rep + 4 * k, 4);
                } else {
                    v = 0;
                    for (int b = 0; 4 * k + b < D; b++) v |= (uint32_t)rep[4 * k + b] << (8 * b);
                }
                diff |= v ^ mine[k];
            }
            if (diff == 0) {
                rep_id = (uint32_t)e;
                break;
            }
        }
        slot = (slot + 1) & E.tab_mask;
        if (probes > E.tab_mask) {  // table full (cannot happen while pool <= cap/2)
            c->failed = 1;
            break;
        }
    }
    // chain hook.  The child that just claimed an empty slot needs none: its id IS the slot's entry, and the next child
    // of this batch to land here finds it there (entry id >= base) when its own exchange returns a stale head — which
    // spares most children (new states are ~85 % on cube3) their second atomic.
    uint32_t prev = NIL;  // the child chained in front of this one (batch-relative index)
    if (!inserted) {
        const uint32_t old_head = atomicExch(&E.tab[slot].head, id);
        if (old_head >= base)
            prev = old_head - base;
        else if (rep_id >= base)
            prev = rep_id - base;  // first to follow the child that inserted the slot in this very batch
    }
    E.child_next[j] = prev;
    E.child_slot[j] = slot;
    E.child_v0[j] = v0;
    E.child_flags[j] = inserted ? F_NEW : 0;  // counted in k_commit (one atomic per block there)
    if (prev != NIL) {  // both ends of the link learn that their chain has company (k_expand cleared the marks)
        E.child_multi[j] = 1;
        E.child_multi[prev] = 1;
    }
}

// dedup B: keep decision in sequential order for child j of the batch (astar.py:81-88 / cpp:247-265): kept iff its
// path cost beats the slot's value before the batch and no earlier child of the batch on the same state has g <= g_j.
// is_min: j is the first occurrence of the batch minimum on its slot (it records the slot's new value);
// first: the sequentially first child on the slot.  A child alone on its slot needs no memory access at all.
struct Dec {
    bool keep, is_min;
    uint32_t first;
};
__device__ __forceinline__ Dec decide_child(const Eng& E, uint32_t base, uint32_t m, uint32_t j, uint32_t slot,
                                            uint32_t v0, uint32_t gj, bool multi) {
    if (!multi) return Dec{gj < v0, true, j};
    bool dominated = false;
    uint32_t gmin = GINF, pmin = NIL, first = j;
    const uint32_t A = (uint32_t)E.A;
    // chain members are children of this batch: indices below m (NIL ends the chain; the step bound is a safety net)
    uint32_t steps = 0;
    for (uint32_t k = E.tab[slot].head - base; k < m && steps <= m; k = E.child_next[k], steps++) {
        const uint32_t gk = E.pop_g[k / A] + 1u;
        dominated |= (k < j && gk <= gj);
        if (gk < gmin || (gk == gmin && k < pmin)) {
            gmin = gk;
            pmin = k;
        }
        first = k < first ? k : first;
    }
    return Dec{(gj < v0) && !dominated, j == pmin, first};
}

// a state first seen in this batch is represented by its sequentially-first child (the node the reference
// inserts, cpp:250) whichever lane won the CAS
__device__ __forceinline__ void fix_representative(const Eng& E, uint32_t base, uint32_t slot, uint32_t id) {
    const uint64_t e = E.tab[slot].entry;
    if ((uint32_t)e >= base && (uint32_t)e != id) E.tab[slot].entry = (e & 0xFFFFFFFF00000000ull) | id;
}

// stand-alone decision launch of the dedup-first ("packed") stepping, where the keep flags are needed before the
// heuristic runs; the fused stepping decides inside k_commit
__global__ __launch_bounds__(256) void k_decide(const Eng* __restrict__ engs) {
    const Eng& E = engs[blockIdx.y];
    Ctl* c = E.ctl;
    if (c->done) return;
    Stamp stamp(E, P_DECIDE);
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    const IterState& S1 = st_next(c);
    const uint32_t m = S1.m, base = S1.base;
    if (j >= m) return;
    const uint32_t slot = E.child_slot[j];
    const bool multi = E.child_multi[j] != 0;
    const uint32_t gj = E.pop_g[j / (uint32_t)E.A] + 1u;
    const Dec d = decide_child(E, base, m, j, slot, E.child_v0[j], gj, multi);
    E.child_flags[j] = (E.child_flags[j] & F_NEW) | (d.keep ? F_KEEP : 0) | (d.is_min ? F_MIN : 0);
    if (multi && j == d.first) fix_representative(E, base, slot, base + j);
}

// ---------------------------------------------------------------------------------------------
// dedup-first stepping: compact the children that survived the CLOSED check into the heuristic batch.
// The reference evaluates the network on every child and only then drops the duplicates (astar.py:272-282);
// a dropped child's heuristic is never used, so evaluating only the kept ones (~85 % on cube3, ~60 % on the
// sliding puzzles) gives the same search.  Row order in the packed batch is arbitrary: kept_pos / pk_src map
// child -> row and row -> child.  One-hot rows are written with the row stride the GEMM wants (tail zero).
// ---------------------------------------------------------------------------------------------
template <int ENV, int DIM, int OH>
__global__ __launch_bounds__(256) void k_pack(const Eng* __restrict__ engs) {
    const Eng& E = engs[blockIdx.y];
    using EV = EnvT<ENV, DIM>;
    Ctl* c = E.ctl;
    if (c->done) {  // the host learns it with the packed row count it reads anyway: pk_n[1] = finished instances, [2] = failed ones
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            atomicAdd(&E.pk_n[1], 1u);
            if (c->failed) atomicAdd(&E.pk_n[2], 1u);
        }
        return;
    }
    Stamp stamp(E, P_PACK);
    const uint32_t m = st_next(c).m, base = st_next(c).base;
    if (blockIdx.x * 256 >= m) return;
    __shared__ uint32_t sh[6];
    __shared__ uint32_t lsrc[256];
    __shared__ __attribute__((aligned(16))) uint8_t lrow[256 * EV::D];
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    const bool keep = j < m && (E.child_flags[j] & F_KEEP);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long mask = __ballot(keep);
    if (lane == 0) sh[wv] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < 4; w++) {
            uint32_t t = sh[w];
            sh[w] = tot;
            tot += t;
        }
        sh[4] = tot;
        sh[5] = tot ? atomicAdd(E.pk_n, tot) : 0u;  // one atomic per 256 children
    }
    __syncthreads();
    const uint32_t total = sh[4], gbase = sh[5];
    const uint32_t lr = sh[wv] + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    if (j < m) E.kept_pos[j] = keep ? gbase + lr : NIL;
    if (keep) lsrc[lr] = j;
    __syncthreads();
    if (total == 0) return;
    if (threadIdx.x < total) E.pk_src[gbase + threadIdx.x] = E.inst * E.M + lsrc[threadIdx.x];
    // network-input bytes of the kept rows (cube3: sticker // 9, cube3.py:77-85; puzzles: the tiles, n_puzzle.py:84-89)
    const uint32_t tb = total * EV::D;
    for (uint32_t q = threadIdx.x; q < tb; q += 256) {
        const uint32_t r = q / EV::D, b = q - r * EV::D;
        const uint32_t v = E.state[(size_t)(base + lsrc[r]) * EV::D + b];
        lrow[q] = (uint8_t)(ENV == DCA_ENV_CUBE3 ? (v * 57u) >> 9 : v);
    }
    __syncthreads();
    uint8_t* gnn = E.pk_nnet + (size_t)gbase * EV::D;
    for (uint32_t q = threadIdx.x; q < tb; q += 256) gnn[q] = lrow[q];
    if constexpr (OH != 0) {
        constexpr uint32_t ROW = EV::D * EV::DEPTH, EPC = 16 / OH;
        const uint32_t cpr = E.pk_stride / EPC;  // 16-byte chunks per packed row
        const uint32_t one16 = E.pk_dtype == DCA_DT_F16 ? 0x3C00u : 0x3F80u;
        uint4* goh = reinterpret_cast<uint4*>(E.pk_onehot + (size_t)gbase * E.pk_stride * OH);
        const uint32_t nch = total * cpr;
        for (uint32_t q = threadIdx.x; q < nch; q += 256) {
            const uint32_t r = q / cpr, e0 = (q - r * cpr) * EPC;
            uint32_t pos = e0 / EV::DEPTH, col = e0 - pos * EV::DEPTH;
            uint32_t nb = pos < (uint32_t)EV::D ? lrow[r * EV::D + pos] : 0xFFFFu;
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
            for (uint32_t k = 0; k < EPC; k++) {
                const bool hot = (e0 + k < ROW) && nb == col;
                if constexpr (OH == 4)
                    w[k] = hot ? 0x3F800000u : 0u;
                else
                    w[k >> 1] |= (hot ? one16 : 0u) << (16 * (k & 1));
                if (++col == EV::DEPTH) {
                    col = 0;
                    ++pos;
                    nb = pos < (uint32_t)EV::D ? lrow[r * EV::D + pos] : 0xFFFFu;
                }
            }
            goh[q] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

// the last workgroup of k_commit to finish closes the iteration (astar.py:317 step_num += 1) and thereby makes the
// state the expansion recorded (S[(iters + 1) & 1]) the current one
// — and leaves what the next iteration's opening launch needs before it can look at FRONT (in all but the rebase
// iterations that launch is k_sel_collect itself, whose workgroups all read these words and none may write them): the
// size of the next batch, and the per-pop words reset.
__device__ __forceinline__ void commit_ticket(const Eng& E, Ctl* c) {
    // (FRONT's size and tombstone count only ever change through returning device-scope atomics, each waited for by its
    // workgroup before the barrier below: the last arrival reads them with device-scope loads — no fence needed, and a
    // release fence per workgroup costs microseconds on the launch's critical path)
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = atomicAdd(&c->ticket_a.v, 1u);
        if (t == gridDim.x - 1) {
            c->ticket_a.v = 0;
            c->iters += 1;
            if (c->stop_after || c->failed) {
                c->done = 1;
            } else {
                const uint32_t fb = c->cur_f;
                const uint32_t live = __hip_atomic_load(&c->open_n[fb].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) -
                                      __hip_atomic_load(&c->front_dead.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - c->front_above;
                c->want = live < (uint32_t)E.B ? live : (uint32_t)E.B;
                c->ret_n.v = 0;
                c->goal_best = ~0ull;
                c->first_solved = NIL;
                c->gbar.v = 0;
                c->grange.kmin = ~0ull;
                c->grange.kmax = 0;
                c->grange.imin = ~0u;
                c->grange.imax = 0;
            }
        }
    }
}

// dedup C: (FUSED: the keep decision,) the new best g per state, push of the kept children (FRONT if key <= T, else BACK)
template <bool FUSED>
__global__ __launch_bounds__(1024) void k_commit(const Eng* __restrict__ engs, int packed) {
    const Eng& E = engs[blockIdx.y];
    Ctl* c = E.ctl;
    if (c->done) return;
    Stamp stamp(E, P_COMMIT);
    __shared__ uint32_t sh[3 * 16 + 3];
    __shared__ uint32_t lh[NBIN];  // this workgroup's pushes into FRONT per selection bin (FRONT's histogram is incremental)
    __shared__ uint64_t red[4][16];
    const IterState& S1 = st_next(c);
    const uint32_t m = S1.m, base = S1.base;
    const uint32_t fb = c->cur_f, bb = c->cur_b;
    const uint64_t T = c->T;
    const uint32_t j = blockIdx.x * 1024 + threadIdx.x;
    if (blockIdx.x * 1024 >= m) {
        commit_ticket(E, c);
        return;
    }
    // The tiers' running key ranges as they stand at the START of the launch, one bound per lane 0-3 (FRONT min / max, BACK min /
    // max), requested now and looked at near the end: a bound is only sent to its atomic if it beats this snapshot (a stale
    // snapshot only means a few atomics that change nothing).  Read at the point of use, the load sat behind the other
    // workgroups' atomics on that very line: 0.5 us of the launch.  Inline asm because the compiler would sink the load to its
    // use; every lane issues it (lanes past 3 re-read lane 0's word: one request per wave) so that the register pair has ONE
    // definition — joined with an initial value the compiler could copy it before the data is there — and it is the wave's
    // oldest load: any later compiler-counted wait for a younger load implies this one has landed.
    uint64_t rng_snap;
    {
        const uint32_t sl = threadIdx.x < 4 ? threadIdx.x : 0u;
        const uint32_t sbuf = sl < 2 ? fb : bb;
        const uint64_t* sp = (sl & 1) ? &c->rng[sbuf].kmax : &c->rng[sbuf].kmin;
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(rng_snap) : "v"(sp) : "memory");
    }
    for (int k = 0; k < kBinsPerThread; k++) lh[kBinsPerThread * threadIdx.x + k] = 0;
    const uint64_t bin_kmin = c->sel_kmin;
    const uint32_t bin_shift = c->shift;
    __syncthreads();
    const bool live = j < m;
    const uint32_t id = base + j;
    bool keep = false, is_new = false;
    uint32_t gj = 0, my_slot = 0;
    // (the heuristic and the solved flag are only needed for kept children, but the keep decision of a chained child is a
    // walk of dependent loads: requested up front, their round trip runs under it instead of behind it)
    float h_early = 0.f;
    uint8_t solved_early = 0;
    if (live) {
        if (!packed) h_early = E.child_h[j];
        solved_early = E.solved[id];
        const uint8_t fl = E.child_flags[j];
        is_new = (fl & F_NEW) != 0;
        gj = E.pop_g[j / (uint32_t)E.A] + 1u;
        const uint32_t slot = E.child_slot[j];
        my_slot = slot;
        bool is_min;
        uint32_t first = j;
        bool multi = false;
        if constexpr (FUSED) {
            multi = E.child_multi[j] != 0;
            const Dec d = decide_child(E, base, m, j, slot, E.child_v0[j], gj, multi);
            keep = d.keep;
            is_min = d.is_min;
            first = d.first;
        } else {
            keep = (fl & F_KEEP) != 0;
            is_min = (fl & F_MIN) != 0;
        }
        if (keep && is_min) {
            E.tab[slot].g = gj;  // one writer per slot: the first occurrence of the batch minimum
            if (E.sem == DCA_SEM_CPP) {
                // cpp:254-257: a shallower duplicate rewrites the CLOSED node's depth/parent/move in place.  A state
                // first seen in this batch is represented by its sequentially-first child (fix_representative may
                // be rewriting the entry right now: both its old and its new id are >= base).
                uint32_t rep = (uint32_t)E.tab[slot].entry;
                if (FUSED && rep >= base) rep = base + first;
                if (rep != id) {
                    E.g[rep] = (int32_t)gj;
                    E.parent[rep] = E.parent[id];
                    E.move[rep] = E.move[id];
                }
            }
        }
        if (FUSED && multi && j == first) fix_representative(E, base, slot, id);
        // the chain marks of this batch have served (k_decide read them in the dedup-first stepping): cleared here, the
        // last launch of the iteration, because the fused expansion cannot clear them itself — another workgroup of that
        // launch may be setting this child's mark at any moment
        E.child_multi[j] = 0;
    }
    uint64_t key = 0;
    uint32_t pid_flag = 0;
    if (keep) {
        // the heuristic is only ever needed for the children that survive the CLOSED check
        const float hraw = packed ? E.pk_h[E.kept_pos[j]] : h_early;
        const float hv = fmaxf(hraw, 0.0f);  // clip_zero (nnet_utils.py:193-194)
        const bool ns = solved_early == 0;
        pid_flag = ns ? 0u : ID_SOLVED;
        double cost;
        if (E.sem == DCA_SEM_PY) {
            // astar.py:196  weights*path_costs + heuristics*logical_not(is_solved), float64, two roundings
            cost = __dadd_rn(__dmul_rn(E.w, (double)gj), __dmul_rn((double)hv, ns ? 1.0 : 0.0));
        } else {
            // cpp:298  values[i]*(!isSolved) + depthPenalty*((float) depth), float32
            cost = (double)__fadd_rn(__fmul_rn(hv, ns ? 1.0f : 0.0f), __fmul_rn(E.wf, (float)gj));
        }
        key = key_of_cost(cost);
    }
    const bool tof = keep && key <= T, tob = keep && key > T;
    const uint32_t hbin = c->hbin;
    if (tof) {
        const uint32_t f = bin_of(key, bin_kmin, bin_shift);
        if (f < hbin) atomicAdd(&lh[f], 1u);
    }
    const uint32_t cnt[3] = {tof ? 1u : 0u, tob ? 1u : 0u, is_new ? 1u : 0u};
    uint32_t* const ctr[3] = {&c->open_n[fb].v, &c->open_n[bb].v, &c->closed_n.v};
    uint32_t pos[3];
    block_reserveK<1024, 3>(cnt, ctr, pos, sh);
    // This workgroup's pushes per selection bin -> FRONT's histogram (block_reserveK's barriers have ordered the LDS counts).
    // In a young search nearly every child lies below the histogram horizon and a workgroup's 1024 children touch 600-900
    // different bins: 235 workgroups x that many device-scope atomics were 10 of this launch's 23 us (iterations 8-27 — the
    // window a 20-step bench episode lives in; tools/engine_probe.py @13=1 times the launch without them, profiles/
    // r05_engine_ab.txt) — and they sat at the very END of the launch, behind everything else.  So: two bins per atomic (a
    // 64-bit add on an aligned pair of 32-bit counters: no carry, counts stay below 2^31), issued HERE, as soon as the counts
    // are complete, so that their trip to the memory side runs under the scattered stores and the range fold below.
    static_assert(kBinsPerThread % 2 == 0, "bins are flushed in aligned pairs");
    if (kBinsPerThread * threadIdx.x < hbin && !(g_tune[13] & 1)) {  // (knob 13 bit 0, diagnostics: timing without the flush)
#pragma unroll
        for (int k = 0; k < kBinsPerThread; k += 2) {
            const uint32_t v0 = lh[kBinsPerThread * threadIdx.x + k], v1 = lh[kBinsPerThread * threadIdx.x + k + 1];
            if (v0 | v1)
                atomicAdd(reinterpret_cast<unsigned long long*>(&E.hist[kBinsPerThread * threadIdx.x + k]),
                          (unsigned long long)v0 | ((unsigned long long)v1 << 32));
        }
    }
    if (is_new && pos[2] < E.max_nodes) E.closed_slots[pos[2]] = my_slot;  // what the next reset clears (k_clear_table_list)
    if (tof) {
        if (pos[0] < E.front_cap) {
            E.open_key[fb][pos[0]] = key;
            E.open_id[fb][pos[0]] = id | pid_flag;
        } else {
            c->failed = 1;
        }
    } else if (tob) {
        if (pos[1] < E.max_nodes) {
            E.open_key[bb][pos[1]] = key;
            E.open_id[bb][pos[1]] = id | pid_flag;
        } else {
            c->failed = 1;
        }
    }
    // running key ranges of the two tiers: ONE atomic per workgroup and bound (a wave-level fold sent thousands of atomics
    // to the same line while OPEN's top cost still rises with every iteration — the first few dozen iterations of a search,
    // exactly the window a short search or a 20-step bench episode lives in: k_commit 21 us there vs 15 us later)
    {
        uint64_t fmn = tof ? key : ~0ull, fmx = tof ? key : 0ull, bmn = tob ? key : ~0ull, bmx = tob ? key : 0ull;
        for (int o = 32; o > 0; o >>= 1) {
            const uint64_t a = __shfl_xor(fmn, o), z = __shfl_xor(fmx, o), x = __shfl_xor(bmn, o), y = __shfl_xor(bmx, o);
            fmn = a < fmn ? a : fmn;
            fmx = z > fmx ? z : fmx;
            bmn = x < bmn ? x : bmn;
            bmx = y > bmx ? y : bmx;
        }
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        if (lane == 0) {
            red[0][wv] = fmn;
            red[1][wv] = fmx;
            red[2][wv] = bmn;
            red[3][wv] = bmx;
        }
        __syncthreads();  // (also: block_reserveK's barriers already ordered the LDS counts; this one covers the early-out threads)
        if (threadIdx.x < 4) {
            const int q = threadIdx.x;
            uint64_t v = red[q][0];
            for (int w = 1; w < 16; w++) v = (q & 1) ? (red[q][w] > v ? red[q][w] : v) : (red[q][w] < v ? red[q][w] : v);
            const uint32_t buf = q < 2 ? fb : bb;
            asm volatile("" : "+v"(rng_snap));  // (the snapshot is the wave's oldest load: long since back)
            if (g_tune[13] & 2) {  // (knob 13 bit 1, diagnostics: timing without the range atomics)
            } else if (q & 1) {
                if (v > rng_snap) atomicMax((unsigned long long*)&c->rng[buf].kmax, (unsigned long long)v);
            } else {
                if (v < rng_snap) atomicMin((unsigned long long*)&c->rng[buf].kmin, (unsigned long long)v);
            }
        }
    }
    commit_ticket(E, c);
}

// per-instance weights of path cost from a device array (dca_engine_set_weights_dev): stream-ordered, no host involvement
// The host setters refuse weights < 0 or NaN (DCA_E_BADARG); a stream-ordered launch cannot refuse, so it clamps them to 0.
__global__ void k_set_weights(Eng* __restrict__ engs, const double* __restrict__ w, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double v = w[i] >= 0.0 ? w[i] : 0.0;  // (NaN fails the comparison)
        engs[i].w = v;
        engs[i].wf = (float)v;
    }
}

// The parents of the last pop (between pop_expand and commit): their state rows, and per parent 1 = popped, 2 = popped and
// solved, 0 = slot unused (short batch, instance finished).  What an ASTAR update backs up (updater.py:44-50: every popped
// node of every instance gets a training target).
__global__ __launch_bounds__(256) void k_gather_popped(const Eng* __restrict__ engs, uint8_t* __restrict__ out_states,
                                                       uint8_t* __restrict__ out_flags) {
    const Eng& E = engs[blockIdx.y];
    const Ctl* c = E.ctl;
    const uint32_t B = (uint32_t)E.B, D = (uint32_t)E.D;
    const uint32_t npop = c->done ? 0u : st_next(c).npop;
    for (uint32_t r = blockIdx.x; r < B; r += gridDim.x) {
        uint8_t* dst = out_states + ((size_t)blockIdx.y * B + r) * D;
        if (r < npop) {
            const uint32_t idf = E.pop_id[r];
            const uint8_t* src = E.state + (size_t)(idf & ID_MASK) * D;
            for (uint32_t b = threadIdx.x; b < D; b += blockDim.x) dst[b] = src[b];
            if (threadIdx.x == 0) out_flags[(size_t)blockIdx.y * B + r] = (idf & ID_SOLVED) ? 2 : 1;
        } else {
            for (uint32_t b = threadIdx.x; b < D; b += blockDim.x) dst[b] = 0;
            if (threadIdx.x == 0) out_flags[(size_t)blockIdx.y * B + r] = 0;
        }
    }
}

__global__ void k_solution(Eng E, int32_t* out /*[0]=len, [1..]=moves root->goal*/, double* path_cost) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Ctl* c = E.ctl;
    out[0] = -1;
    *path_cost = 0.0;
    uint32_t n;
    if (E.sem == DCA_SEM_PY) {
        if (c->goal_best == ~0ull && !(c->done && c->stop_after)) return;
        n = c->goal_id;
        *path_cost = (double)E.g[n];  // astar.py:229 node.path_cost
    } else {
        if (!st_cur(c).has_best) return;
        n = st_cur(c).best_id;
    }
    int len = 0;
    // astar.py:218-223 walks parents to the root; cpp:337-341 walks while depth > 0
    while ((E.sem == DCA_SEM_PY ? E.parent[n] != NIL : E.g[n] > 0) && len < kMaxMoves - 1) {
        out[1 + len++] = E.move[n];
        n = E.parent[n];
    }
    for (int i = 0; i < len / 2; i++) {
        int32_t tmp = out[1 + i];
        out[1 + i] = out[len - i];
        out[len - i] = tmp;
    }
    out[0] = len;
    if (E.sem == DCA_SEM_CPP) *path_cost = (double)len;  // astar.py:554 sum of unit transition costs
}


}  // namespace dca


using namespace dca;

// ---------------------------------------------------------------------------------------------
// host side.  One dca_engine = K independent search instances of the same geometry that share every launch
// (grid.y = instance): a batch-20 000 iteration is launch/latency bound, so stepping K scrambles together costs little
// more than stepping one (the reference's AStar also steps a list of instances together, astar.py:232-317).
// ---------------------------------------------------------------------------------------------
constexpr int kMaxInstances = 64;
constexpr int kGraphSlots = 48, kGraphChunk = 64;

struct dca_engine {
    int K;
    Eng E[kMaxInstances];  // host copies; E[i].ctl etc. are device pointers
    Eng* d_engs;           // device copy of E[0..K)
    uint8_t* nnet_all;     // [K*M, D]   network-input rows of all instances, contiguous
    uint8_t* onehot_all;   // [K*M, D*depth] or null
    float* h_all;          // [K*M]
    Ctl* h_ctl;            // pinned
    int32_t* h_moves;
    double* d_cost;
    double* h_cost;
    uint8_t* h_stage;
    uint32_t* pk_n;        // packed stepping: device row counter, its pinned host mirror, the packed buffers
    uint32_t* h_pk_n;
    uint8_t *pk_nnet, *pk_onehot;
    uint32_t* pk_src;
    float* pk_h;
    int64_t pk_rows;       // rows packed by the last dca_engine_pop_expand_packed
    int phase;             // 0 idle, 1 between pop_expand and commit, 2 between pop_expand_packed and commit_packed
    bool w_dev;            // the device copy holds weights the host mirror has not seen (dca_engine_set_weights_dev)
    uint8_t tab_cleared[kMaxInstances];  // the instance's CLOSED table has been cleared in full at least once
    unsigned collect_blocks;  // grid of k_sel_collect: two workgroups per CU, all resident (its giant-bin path barriers across it)
    long collect_resident;    // workgroups of k_sel_collect the device can hold at once per the occupancy query (-1: query failed)
    // run_builtin's replayable chunks: a chunk of n <= kGraphChunk consecutive iterations is one hipGraph, keyed by its
    // pattern of rebase iterations (bit i = iteration i of the chunk runs the refill check) — the boundary between two
    // graph launches costs ~6 us on the device, the one between two launches inside a graph ~2
    struct GraphSlot {
        unsigned long long mask;
        int n;
        hipGraph_t graph;
        hipGraphExec_t exec;
    } gslot[kGraphSlots];
    int gslot_next;  // round-robin victim once every slot is taken
    int graph_heur;
    long host_iter;        // iterations enqueued since the last reset of instance 0 (drives the refill cadence)
    unsigned long long* d_prof;  // [P_COUNT][kProfSlots][2] device wall-clock stamps of the profiled launches
    unsigned long long* h_prof;  // pinned mirror
    void* allocs[kMaxInstances * 48 + 16];
    int nalloc;
};

namespace {

template <typename T>
int dev_alloc(dca_engine* e, T** p, size_t count) {
    if (e->nalloc >= (int)(sizeof(e->allocs) / sizeof(e->allocs[0]))) {
        set_error("allocation table full");
        return DCA_E_NOMEM;
    }
    void* q = nullptr;
    hipError_t err = hipMalloc(&q, count * sizeof(T) + 256);
    if (err != hipSuccess) {
        set_error("hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(err));
        return DCA_E_NOMEM;
    }
    e->allocs[e->nalloc++] = q;
    *p = reinterpret_cast<T*>(q);
    return 0;
}

constexpr int kScanGrid = kScanBlocks;
// Rebase iterations: every kRefillPeriod-th — and each of the first kRampIters after a reset.  A search starts from one
// entry, so the binning taken at iteration 0 (key range = the root's key) puts every child of the next iterations into
// the last bin: one bin of 10^5..10^6 entries for k_rank to order on a single workgroup (1-2 ms per iteration, measured:
// k_rank's rocprofv3 maximum).  While OPEN is still growing by a factor of A per iteration a fresh binning each time is cheap.
constexpr int kRampIters = 8;
static inline bool rebase_due(long host_iter) { return host_iter < kRampIters || host_iter % kRefillPeriod == 0; }
static int h_tune[16];  // host copy of the diagnostic knobs (dca_debug_tune)

inline dim3 gxy(unsigned x, const dca_engine* e) { return dim3(x, (unsigned)e->K); }
// the CLOSED probe runs inside the expansion launch (four launches per iteration); knob 11 restores the separate k_probe
// launch (round-3 behaviour, the A/B reference)
inline bool fuse_probe() { return h_tune[11] == 0; }

template <int ENV, int DIM>
int launch_expand_env(const dca_engine* e, int heur_id, bool want_oh, bool want_nn, hipStream_t s) {
    using TL = Tile<ENV, DIM, kEngTile>;
    const Eng& E = e->E[0];
    dim3 g = gxy((E.B + kEngTile - 1) / kEngTile, e), b(kThreads);
    // tile + tables, then the staged child rows (16 parents x A children x D bytes)
    const size_t lds = TL::LDS_BYTES + ((kEngTile * EnvT<ENV, DIM>::A * EnvT<ENV, DIM>::D + 15) / 16) * 16 + 64;  // (+ slack: the one-hot loop peeks one byte past the tile)
    const bool fuse = fuse_probe();
    const int wnn = ((want_nn && heur_id < 0) || h_tune[12] != 0) ? 1 : 0;  // (knob 12: always write them, the round-4 behaviour, for A/B runs)
    if (E.onehot == nullptr || !want_oh) {
        if (fuse)
            hipLaunchKernelGGL((k_expand<ENV, DIM, 0, true>), g, b, lds, s, e->d_engs, heur_id, wnn);
        else
            hipLaunchKernelGGL((k_expand<ENV, DIM, 0, false>), g, b, lds, s, e->d_engs, heur_id, wnn);
    } else if (E.oh_dtype == DCA_DT_F32) {
        if (fuse)
            hipLaunchKernelGGL((k_expand<ENV, DIM, 4, true>), g, b, lds, s, e->d_engs, heur_id, wnn);
        else
            hipLaunchKernelGGL((k_expand<ENV, DIM, 4, false>), g, b, lds, s, e->d_engs, heur_id, wnn);
    } else {
        if (fuse)
            hipLaunchKernelGGL((k_expand<ENV, DIM, 2, true>), g, b, lds, s, e->d_engs, heur_id, wnn);
        else
            hipLaunchKernelGGL((k_expand<ENV, DIM, 2, false>), g, b, lds, s, e->d_engs, heur_id, wnn);
    }
    return launch_check("k_expand");
}

int launch_expand(const dca_engine* e, int heur_id, bool want_oh, bool want_nn, hipStream_t s) {
    const Eng& E = e->E[0];
    if (E.env == DCA_ENV_CUBE3) return launch_expand_env<DCA_ENV_CUBE3, 0>(e, heur_id, want_oh, want_nn, s);
    if (E.env == DCA_ENV_LIGHTSOUT) return launch_expand_env<DCA_ENV_LIGHTSOUT, 7>(e, heur_id, want_oh, want_nn, s);
    switch (E.dim) {
        case 4: return launch_expand_env<DCA_ENV_NPUZZLE, 4>(e, heur_id, want_oh, want_nn, s);
        case 5: return launch_expand_env<DCA_ENV_NPUZZLE, 5>(e, heur_id, want_oh, want_nn, s);
        case 6: return launch_expand_env<DCA_ENV_NPUZZLE, 6>(e, heur_id, want_oh, want_nn, s);
        case 7: return launch_expand_env<DCA_ENV_NPUZZLE, 7>(e, heur_id, want_oh, want_nn, s);
    }
    return DCA_E_BADARG;
}

template <int ENV, int DIM>
int launch_pack_env(const dca_engine* e, hipStream_t s) {
    const Eng& E = e->E[0];
    const dim3 g = gxy((E.M + 255) / 256, e), b(256);
    if (E.pk_onehot == nullptr)
        hipLaunchKernelGGL((k_pack<ENV, DIM, 0>), g, b, 0, s, e->d_engs);
    else if (E.pk_dtype == DCA_DT_F32)
        hipLaunchKernelGGL((k_pack<ENV, DIM, 4>), g, b, 0, s, e->d_engs);
    else
        hipLaunchKernelGGL((k_pack<ENV, DIM, 2>), g, b, 0, s, e->d_engs);
    return launch_check("k_pack");
}

int launch_pack(const dca_engine* e, hipStream_t s) {
    const Eng& E = e->E[0];
    if (E.env == DCA_ENV_CUBE3) return launch_pack_env<DCA_ENV_CUBE3, 0>(e, s);
    if (E.env == DCA_ENV_LIGHTSOUT) return launch_pack_env<DCA_ENV_LIGHTSOUT, 7>(e, s);
    switch (E.dim) {
        case 4: return launch_pack_env<DCA_ENV_NPUZZLE, 4>(e, s);
        case 5: return launch_pack_env<DCA_ENV_NPUZZLE, 5>(e, s);
        case 6: return launch_pack_env<DCA_ENV_NPUZZLE, 6>(e, s);
        case 7: return launch_pack_env<DCA_ENV_NPUZZLE, 7>(e, s);
    }
    return DCA_E_BADARG;
}

void launch_probe(const dca_engine* e, hipStream_t s) {
    const Eng& E = e->E[0];
    const dim3 g = gxy((E.M + 255) / 256, e), b(256);
    switch (E.D) {
        case 54: hipLaunchKernelGGL(k_probe<54>, g, b, 0, s, e->d_engs); break;
        case 16: hipLaunchKernelGGL(k_probe<16>, g, b, 0, s, e->d_engs); break;
        case 25: hipLaunchKernelGGL(k_probe<25>, g, b, 0, s, e->d_engs); break;
        case 36: hipLaunchKernelGGL(k_probe<36>, g, b, 0, s, e->d_engs); break;
        default: hipLaunchKernelGGL(k_probe<49>, g, b, 0, s, e->d_engs); break;
    }
}

constexpr size_t kRankLdsBytes = (size_t)kLdsEnt * 12;

int enqueue_first_half(dca_engine* e, int heur_id, bool with_refill, hipStream_t s, bool want_oh = true, bool want_nn = true) {
    const Eng* d = e->d_engs;
    if (with_refill) {
        hipLaunchKernelGGL(k_refill_hist, gxy(kScanGrid, e), dim3(256), 0, s, d);
        hipLaunchKernelGGL(k_refill_scan, gxy(1, e), dim3(1024), 0, s, d);
        hipLaunchKernelGGL(k_refill_move, gxy(kScanGrid, e), dim3(256), 0, s, d);
    }
    // FRONT's selection histogram is recounted only in the refill-check ("rebase") iterations — every kRefillPeriod-th —
    // and maintained incrementally in between (k_sel_scan's writeback + k_commit's pushes)
    if (with_refill) hipLaunchKernelGGL(k_front_rebase, gxy(kCollectBlocks, e), dim3(256), 0, s, d);
    // the iteration's opening launch: k_sel_scan in a rebase iteration, k_sel_collect itself otherwise (FUSED)
    if (with_refill || h_tune[6] != 0) {  // (knob 6: a k_sel_scan launch in every iteration, round-2 behaviour)
        hipLaunchKernelGGL(k_sel_scan, gxy(1, e), dim3(1024), 0, s, d, with_refill ? 1 : 0);
        hipLaunchKernelGGL(k_sel_collect<false>, gxy(e->collect_blocks, e), dim3(256), sizeof(CollectLds), s, d);
    } else {
        hipLaunchKernelGGL(k_sel_collect<true>, gxy(e->collect_blocks, e), dim3(256), sizeof(CollectLds), s, d);
    }
    hipLaunchKernelGGL(k_rank, gxy(kRankBlocks, e), dim3(RT), kRankLdsBytes, s, d);
    if (int rc = launch_check("select kernels")) return rc;
    return launch_expand(e, heur_id, want_oh, want_nn, s);
}

// CLOSED check of the batch's children.  with_decide: the stand-alone keep decision of the dedup-first stepping
// (k_pack needs the flags before the heuristic runs); otherwise k_commit<true> decides while it pushes.
int enqueue_dedup(dca_engine* e, bool with_decide, hipStream_t s) {
    const Eng& E = e->E[0];
    if (!fuse_probe()) launch_probe(e, s);
    if (with_decide) hipLaunchKernelGGL(k_decide, gxy((E.M + 255) / 256, e), dim3(256), 0, s, e->d_engs);
    return launch_check("dedup kernels");
}

int enqueue_commit(dca_engine* e, bool packed, hipStream_t s) {
    const Eng& E = e->E[0];
    const dim3 g = gxy((E.M + 1023) / 1024, e);
    if (packed)
        hipLaunchKernelGGL(k_commit<false>, g, dim3(1024), 0, s, e->d_engs, 1);
    else
        hipLaunchKernelGGL(k_commit<true>, g, dim3(1024), 0, s, e->d_engs, 0);
    return launch_check("k_commit");
}

int enqueue_second_half(dca_engine* e, hipStream_t s) {
    if (int rc = enqueue_dedup(e, false, s)) return rc;
    return enqueue_commit(e, false, s);
}

void drop_graph_slot(dca_engine::GraphSlot& g) {
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (g.graph) (void)hipGraphDestroy(g.graph);
    g.exec = nullptr;
    g.graph = nullptr;
    g.n = 0;
    g.mask = 0;
}

void drop_graphs(dca_engine* e) {
    for (int g = 0; g < kGraphSlots; g++) drop_graph_slot(e->gslot[g]);
    e->gslot_next = 0;
}

// dca_engine_set_weights_dev writes w / wf of the DEVICE copy only (stream-ordered, no host involvement: ADVICE r05).  Before
// the host mirror is used as a whole — any upload, any host-side weight setter — the device's weights are read back into it, so
// that a later upload cannot revert them.  (Kernels that take an Eng by value from the mirror read w only as w * 0 at the root.)
int pull_dev_weights(dca_engine* e) {
    if (!e->w_dev) return 0;
    DCA_HIP(hipDeviceSynchronize());
    std::vector<Eng> tmp((size_t)e->K);
    DCA_HIP(hipMemcpy(tmp.data(), e->d_engs, sizeof(Eng) * (size_t)e->K, hipMemcpyDeviceToHost));
    for (int i = 0; i < e->K; i++) {
        e->E[i].w = tmp[(size_t)i].w;
        e->E[i].wf = tmp[(size_t)i].wf;
    }
    e->w_dev = false;
    return 0;
}

int upload_engs(dca_engine* e) {
    if (int rc = pull_dev_weights(e)) return rc;
    DCA_HIP(hipMemcpy(e->d_engs, e->E, sizeof(Eng) * (size_t)e->K, hipMemcpyHostToDevice));
    return 0;
}

int fetch_ctl(dca_engine* e, int inst, hipStream_t s) {
    DCA_HIP(hipMemcpyAsync(e->h_ctl, e->E[inst].ctl, sizeof(Ctl), hipMemcpyDeviceToHost, s));
    DCA_HIP(hipStreamSynchronize(s));
    return 0;
}

#define DCA_INST(e, i)                                                        \
    do {                                                                      \
        DCA_ARG((e) != nullptr);                                              \
        if ((i) < 0 || (i) >= (e)->K) {                                       \
            set_error("instance %d out of range (engine has %d)", (i), (e)->K); \
            return DCA_E_BADARG;                                              \
        }                                                                     \
    } while (0)

}  // namespace

extern "C" {

int dca_engine_create_multi(dca_engine** out, int env, int dim, double weight, int batch_size, int64_t max_nodes,
                            int semantics, int onehot_dtype, int num_instances) {
    DCA_ARG(out != nullptr);
    *out = nullptr;
    DCA_ARG(env == DCA_ENV_CUBE3 || (env == DCA_ENV_NPUZZLE && dim >= 4 && dim <= 7) || (env == DCA_ENV_LIGHTSOUT && dim == 7));
    DCA_ARG(batch_size >= 1 && batch_size <= (1 << 22));
    DCA_ARG(semantics == DCA_SEM_PY || semantics == DCA_SEM_CPP);
    DCA_ARG(onehot_dtype >= -1 && onehot_dtype <= DCA_DT_BF16);
    DCA_ARG(weight >= 0.0);
    DCA_ARG(num_instances >= 1 && num_instances <= kMaxInstances);
    const int D = env == DCA_ENV_CUBE3 ? 54 : dim * dim;
    const int A = env == DCA_ENV_CUBE3 ? 12 : env == DCA_ENV_LIGHTSOUT ? D : 4;
    const int depth = env == DCA_ENV_NPUZZLE ? D : 6;
    const int64_t Mll = (int64_t)batch_size * A;
    DCA_ARG(max_nodes >= Mll + 16 && max_nodes <= 0x7FFFFF00ll);
    // CLOSED slots: the power of two at or above 2 per node id.  (A 256 MiB instead of a 512 MiB table at the bench shape —
    // the size of the Infinity Cache — changes nothing: k_expand 28.1 us either way, profiles/r05_engine_ab.txt.  The probe is
    // bound by the memory-side atomics' rate, not by where the slots live.)
    uint64_t cap = 1024;
    while (cap < 2ull * (uint64_t)max_nodes) cap <<= 1;
    if (cap > 0x80000000ull) {
        set_error("max_nodes too large for a 32-bit slot index");
        return DCA_E_BADARG;
    }
    dca_engine* e = new (std::nothrow) dca_engine();
    if (!e) return DCA_E_NOMEM;
    memset(e, 0, sizeof(*e));
    e->K = num_instances;
    {
        // k_sel_collect's giant-bin path runs grid barriers: every workgroup must be resident.  Two 256-thread workgroups
        // (74 KB of LDS each) fit a CU; a device that exposes fewer CUs (partition modes) gets a smaller grid.
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        e->collect_blocks = (unsigned)(2 * cus < kCollectBlocks ? 2 * cus : kCollectBlocks);
        if (h_tune[4] > 0 && h_tune[4] < (int)e->collect_blocks) e->collect_blocks = (unsigned)h_tune[4];
        // K instances share every launch (grid.y): keep the launch at about two workgroups per CU in total
        if (num_instances > 1) {
            const unsigned per = e->collect_blocks / (unsigned)num_instances;
            e->collect_blocks = per < 8u ? 8u : per;
        }
    }
    const size_t K = (size_t)num_instances;
    const size_t N = (size_t)max_nodes, M = (size_t)Mll, Bz = (size_t)batch_size + 64;
    int rc = 0;
    // batch buffers shared by all instances (contiguous, so one heuristic call serves every instance)
    rc = dev_alloc(e, &e->nnet_all, K * M * D + 64);
    if (!rc) rc = dev_alloc(e, &e->h_all, K * M);
    if (!rc && onehot_dtype >= 0) rc = dev_alloc(e, &e->onehot_all, K * M * D * depth * (onehot_dtype == DCA_DT_F32 ? 4 : 2) + 64);
    if (!rc) rc = dev_alloc(e, &e->d_engs, K);
    if (!rc) rc = dev_alloc(e, &e->d_cost, 1);
    for (size_t i = 0; i < K && !rc; i++) {
        Eng& E = e->E[i];
        E.env = env;
        E.dim = dim;
        E.D = D;
        E.A = A;
        E.B = batch_size;
        E.sem = semantics;
        E.oh_dtype = onehot_dtype;
        E.depth = depth;
        E.w = weight;
        E.wf = (float)weight;  // cpp:353 (float) atof(argv[2])
        E.max_nodes = (uint32_t)max_nodes;
        E.M = (uint32_t)Mll;
        E.coop = (num_instances == 1 && h_tune[5] == 0) ? 1 : 0;  // (knob 5: giant-bin path off, round-2 behaviour)
        E.f_keep = (uint32_t)(32 * batch_size > 65536 ? 32 * batch_size : 65536);
        E.f_max = 3 * E.f_keep;
        E.tab_cap = (uint32_t)cap;
        E.tab_mask = (uint32_t)(cap - 1);
        E.nnet_in = e->nnet_all + i * M * D;
        E.child_h = e->h_all + i * M;
        E.onehot = e->onehot_all ? e->onehot_all + i * M * D * depth * (onehot_dtype == DCA_DT_F32 ? 4 : 2) : nullptr;
#define ALLOC(field, count) \
    if (!rc) rc = dev_alloc(e, &E.field, (count))
        ALLOC(state, N * D + 64);
        ALLOC(g, N);
        ALLOC(parent, N);
        ALLOC(move, N);
        ALLOC(solved, N);
        ALLOC(tab, (size_t)cap);
        ALLOC(closed_slots, N);
        ALLOC(coop_off, 16);
        E.front_cap = (uint32_t)(N + (size_t)(2 * kRefillPeriod) * Bz);
        for (int b = 0; b < 4; b++) {  // FRONT and its compaction target (0/1) + BACK and its compaction target (2/3)
            ALLOC(open_key[b], b < 2 ? (size_t)E.front_cap : N);
            ALLOC(open_id[b], b < 2 ? (size_t)E.front_cap : N);
        }
        ALLOC(hist, NBIN);
        ALLOC(rhist, NBIN);
        ALLOC(pre, kSegs + 16);
        ALLOC(fill, kSegs + 8);
        ALLOC(subhist, (size_t)kMaxLevels * kSub);
        ALLOC(part, 4 * 1024);
        ALLOC(tmp_key, N);
        ALLOC(tmp_id, N);
        ALLOC(tmp_f, N);
        ALLOC(tmp_idx, N);
        ALLOC(ord_key, N);
        ALLOC(ord_id, N);
        ALLOC(big_list, 8 * kSegs);
        ALLOC(pop_key, Bz);
        ALLOC(pop_id, Bz);
        ALLOC(pop_g, Bz);
        ALLOC(child_hash, M);
        ALLOC(child_slot, M);
        ALLOC(child_next, M);
        ALLOC(child_flags, M);
        ALLOC(child_v0, M);
        ALLOC(child_multi, M);
        ALLOC(root_nnet, 64);
        ALLOC(d_moves, kMaxMoves);
        ALLOC(ctl, 1);
#undef ALLOC
        if (!rc) {
            (void)hipMemset(E.hist, 0, NBIN * sizeof(uint32_t));
            (void)hipMemset(E.rhist, 0, NBIN * sizeof(uint32_t));
            (void)hipMemset(E.fill, 0, (kSegs + 8) * sizeof(uint32_t));
            (void)hipMemset(E.subhist, 0, (size_t)kMaxLevels * kSub * sizeof(uint32_t));
            (void)hipMemset(E.child_multi, 0, M);
            (void)hipMemset(E.coop_off, 0, 16 * sizeof(uint32_t));
            (void)hipMemset(E.ctl, 0, sizeof(Ctl));
        }
    }
    if (!rc) {
        hipError_t err = hipHostMalloc((void**)&e->h_ctl, sizeof(Ctl) + kMaxMoves * sizeof(int32_t) + 512);
        if (err != hipSuccess) rc = hip_fail(err, "hipHostMalloc");
    }
    if (!rc) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sel_collect<false>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(CollectLds));
        if (err == hipSuccess)
            err = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sel_collect<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(CollectLds));
        if (err != hipSuccess) rc = hip_fail(err, "hipFuncSetAttribute(k_sel_collect)");
        // The giant-bin path barriers across k_sel_collect's grid: ask the runtime how many of its workgroups a CU holds and
        // keep the path only if the whole grid fits (the query is advisory — it cannot see other processes or streams — so
        // the first barrier of every giant iteration checks residency for real and falls back by itself, collect_grid_barrier).
        if (err == hipSuccess) {
            int per_cu_a = 0, per_cu_b = 0, dev = 0, cus = 0;
            (void)hipGetDevice(&dev);
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            hipError_t oa = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_a, k_sel_collect<false>, 256, sizeof(CollectLds));
            hipError_t ob = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_b, k_sel_collect<true>, 256, sizeof(CollectLds));
            const int per_cu = per_cu_a < per_cu_b ? per_cu_a : per_cu_b;
            e->collect_resident = (oa == hipSuccess && ob == hipSuccess && cus > 0) ? (long)per_cu * cus : -1;
            if (e->collect_resident >= 0 && e->collect_resident < (long)e->collect_blocks) {
                for (size_t i = 0; i < K; i++) e->E[i].coop = 0;
            }
            (void)hipGetLastError();
        }
    }
    if (!rc) {
        // k_rank buckets a bin inside 96 KB of dynamic LDS: beyond the default limit
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(k_rank),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRankLdsBytes);
        if (err != hipSuccess) rc = hip_fail(err, "hipFuncSetAttribute(k_rank)");
    }
    if (rc) {
        dca_engine_destroy(e);
        return rc;
    }
    e->h_moves = reinterpret_cast<int32_t*>(reinterpret_cast<uint8_t*>(e->h_ctl) + ((sizeof(Ctl) + 15) & ~15ul));
    e->h_cost = reinterpret_cast<double*>(e->h_moves + kMaxMoves);
    e->h_stage = reinterpret_cast<uint8_t*>(e->h_cost + 1);
    (void)hipMemset(e->nnet_all, 0, K * M * D);
    if (int urc = upload_engs(e)) {
        dca_engine_destroy(e);
        return urc;
    }
    // every instance starts "done" until it is reset with a root (so unused instances are inert)
    for (size_t i = 0; i < K; i++) {
        int32_t one = 1;
        (void)hipMemcpy(&e->E[i].ctl->done, &one, sizeof(one), hipMemcpyHostToDevice);
    }
    *out = e;
    return 0;
}

int dca_engine_create(dca_engine** out, int env, int dim, double weight, int batch_size, int64_t max_nodes,
                      int semantics, int onehot_dtype) {
    return dca_engine_create_multi(out, env, dim, weight, batch_size, max_nodes, semantics, onehot_dtype, 1);
}

int dca_engine_num_instances(dca_engine* e) { return e ? e->K : 0; }

void dca_engine_destroy(dca_engine* e) {
    if (!e) return;
    drop_graphs(e);
    for (int i = 0; i < e->nalloc; i++) (void)hipFree(e->allocs[i]);
    if (e->h_ctl) (void)hipHostFree(e->h_ctl);
    if (e->h_pk_n) (void)hipHostFree(e->h_pk_n);
    if (e->h_prof) (void)hipHostFree(e->h_prof);
    delete e;
}

int dca_engine_reset_instance(dca_engine* e, int inst, const uint8_t* root, void* stream) {
    DCA_INST(e, inst);
    DCA_ARG(root != nullptr);
    hipStream_t s = (hipStream_t)stream;
    Eng& E = e->E[inst];
    for (int i = 0; i < E.D; i++) DCA_ARG(root[i] < (E.env == DCA_ENV_CUBE3 ? 54 : E.env == DCA_ENV_LIGHTSOUT ? 2 : E.D));
    // the staging block is reused: make sure an earlier reset's copy has left it
    DCA_HIP(hipStreamSynchronize(s));
    memcpy(e->h_stage, root, (size_t)E.D);
    DCA_HIP(hipMemcpyAsync(E.state, e->h_stage, (size_t)E.D, hipMemcpyHostToDevice, s));
    {
        // An iteration abandoned between its two halves has claimed slots that no instance's list holds yet (k_commit
        // records them): EVERY instance of the engine took part in that launch, so every one of them owes a full clear at
        // its next reset — not only the one that happens to be reset first.
        if (e->phase != 0)
            for (int i = 0; i < e->K; i++) e->tab_cleared[i] = 0;
        const int force = e->tab_cleared[inst] == 0 ? 1 : 0;
        hipLaunchKernelGGL(k_init_table, dim3(4096), dim3(256), 0, s, E.tab, E.tab_cap, E.ctl, force);
        hipLaunchKernelGGL(k_clear_table_list, dim3(1024), dim3(256), 0, s, E.tab, E.tab_cap, E.closed_slots, E.ctl, force);
        e->tab_cleared[inst] = 1;
    }
    hipLaunchKernelGGL(k_reset, dim3(1), dim3(64), 0, s, E);
    DCA_HIP(hipMemsetAsync(E.child_multi, 0, (size_t)E.M, s));  // chain marks of an abandoned iteration (k_commit clears its own)
    e->phase = 0;
    e->host_iter = 0;  // the next iteration is a rebase iteration (full histogram) for every instance
    return launch_check("k_reset");
}
int dca_engine_reset(dca_engine* e, const uint8_t* root, void* stream) {
    return dca_engine_reset_instance(e, 0, root, stream);
}

int dca_engine_root_commit_instance(dca_engine* e, int inst, const float* h_root, void* stream) {
    DCA_INST(e, inst);
    if (e->E[inst].sem != DCA_SEM_PY) return 0;
    DCA_ARG(h_root != nullptr);
    hipLaunchKernelGGL(k_root_commit, dim3(1), dim3(64), 0, (hipStream_t)stream, e->E[inst], h_root);
    return launch_check("k_root_commit");
}
int dca_engine_root_commit(dca_engine* e, const float* h_root, void* stream) {
    return dca_engine_root_commit_instance(e, 0, h_root, stream);
}

int dca_engine_root_nnet_in_instance(dca_engine* e, int inst, const uint8_t** nnet_in) {
    DCA_INST(e, inst);
    DCA_ARG(nnet_in != nullptr);
    *nnet_in = e->E[inst].root_nnet;
    return 0;
}
int dca_engine_root_nnet_in(dca_engine* e, const uint8_t** nnet_in) {
    return dca_engine_root_nnet_in_instance(e, 0, nnet_in);
}

int dca_engine_pop_expand(dca_engine* e, const uint8_t** nnet_in, const void** onehot, int64_t* m_capacity,
                          void* stream) {
    DCA_ARG(e != nullptr);
    if (e->phase != 0) {
        set_error("dca_engine_pop_expand called twice without dca_engine_commit");
        return DCA_E_STATE;
    }
    if (int rc = enqueue_first_half(e, -1, rebase_due(e->host_iter++), (hipStream_t)stream)) return rc;
    if (nnet_in) *nnet_in = e->nnet_all;
    if (onehot) *onehot = e->onehot_all;
    if (m_capacity) *m_capacity = (int64_t)e->E[0].M * e->K;
    e->phase = 1;
    return 0;
}

int dca_engine_commit(dca_engine* e, const float* h, void* stream) {
    DCA_ARG(e != nullptr && h != nullptr);
    if (e->phase != 1) {
        set_error("dca_engine_commit without a preceding dca_engine_pop_expand");
        return DCA_E_STATE;
    }
    hipStream_t s = (hipStream_t)stream;
    DCA_HIP(hipMemcpyAsync(e->h_all, h, (size_t)e->E[0].M * e->K * sizeof(float), hipMemcpyDeviceToDevice, s));
    e->phase = 0;
    return enqueue_second_half(e, s);
}


int dca_engine_enable_packed(dca_engine* e, int onehot_dtype, int64_t onehot_row_stride) {
    DCA_ARG(e != nullptr && onehot_dtype >= -1 && onehot_dtype <= DCA_DT_BF16);
    if (e->pk_n != nullptr) {
        set_error("dca_engine_enable_packed called twice");
        return DCA_E_STATE;
    }
    const Eng& E0 = e->E[0];
    const int64_t row = (int64_t)E0.D * E0.depth;
    const int esz = onehot_dtype == DCA_DT_F32 ? 4 : 2;
    if (onehot_dtype >= 0) DCA_ARG(onehot_row_stride >= row && (onehot_row_stride * esz) % 16 == 0 && onehot_row_stride < (1 << 20));
    const size_t rows = (((size_t)E0.M * e->K + 1023) / 1024) * 1024;  // callers may round the batch up to 1024 rows
    if (int rc = dev_alloc(e, &e->pk_n, 64)) return rc;
    if (int rc = dev_alloc(e, &e->pk_src, rows)) return rc;
    if (int rc = dev_alloc(e, &e->pk_nnet, rows * E0.D)) return rc;
    if (int rc = dev_alloc(e, &e->pk_h, rows)) return rc;
    DCA_HIP(hipMemset(e->pk_nnet, 0, rows * E0.D));
    if (onehot_dtype >= 0) {
        if (int rc = dev_alloc(e, &e->pk_onehot, rows * (size_t)onehot_row_stride * esz)) return rc;
        DCA_HIP(hipMemset(e->pk_onehot, 0, rows * (size_t)onehot_row_stride * esz));
    }
    DCA_HIP(hipHostMalloc((void**)&e->h_pk_n, 64, hipHostMallocDefault));
    for (int i = 0; i < e->K; i++) {
        Eng& E = e->E[i];
        if (int rc = dev_alloc(e, &E.kept_pos, (size_t)E.M)) return rc;
        E.pk_n = e->pk_n;
        E.pk_src = e->pk_src;
        E.pk_nnet = e->pk_nnet;
        E.pk_onehot = e->pk_onehot;
        E.pk_h = e->pk_h;
        E.pk_stride = (uint32_t)(onehot_dtype >= 0 ? onehot_row_stride : 0);
        E.pk_dtype = onehot_dtype;
        E.inst = (uint32_t)i;
    }
    return upload_engs(e);
}

int dca_engine_pop_expand_packed(dca_engine* e, const uint8_t** nnet_in, const void** onehot, const uint32_t** src,
                                 int64_t* rows, void* stream) {
    DCA_ARG(e != nullptr && rows != nullptr);
    if (e->pk_n == nullptr) {
        set_error("dca_engine_pop_expand_packed before dca_engine_enable_packed");
        return DCA_E_STATE;
    }
    if (e->phase != 0) {
        set_error("dca_engine_pop_expand_packed called twice without dca_engine_commit_packed");
        return DCA_E_STATE;
    }
    hipStream_t s = (hipStream_t)stream;
    DCA_HIP(hipMemsetAsync(e->pk_n, 0, 4 * sizeof(uint32_t), s));
    if (int rc = enqueue_first_half(e, -1, rebase_due(e->host_iter++), s, false, false)) return rc;
    if (int rc = enqueue_dedup(e, true, s)) return rc;
    if (int rc = launch_pack(e, s)) return rc;
    DCA_HIP(hipMemcpyAsync(e->h_pk_n, e->pk_n, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    DCA_HIP(hipStreamSynchronize(s));
    e->pk_rows = (int64_t)*e->h_pk_n;
    *rows = e->pk_rows;
    if (nnet_in) *nnet_in = e->pk_nnet;
    if (onehot) *onehot = e->pk_onehot;
    if (src) *src = e->pk_src;
    e->phase = 2;
    return 0;
}

int dca_engine_packed_state(dca_engine* e, int* instances_done, int* instances_failed) {
    DCA_ARG(e != nullptr);
    if (e->h_pk_n == nullptr) {
        set_error("dca_engine_packed_state before dca_engine_enable_packed");
        return DCA_E_STATE;
    }
    if (instances_done) *instances_done = (int)e->h_pk_n[1];
    if (instances_failed) *instances_failed = (int)e->h_pk_n[2];
    return 0;
}

int dca_engine_commit_packed(dca_engine* e, const float* h, void* stream) {
    DCA_ARG(e != nullptr);
    if (e->phase != 2) {
        set_error("dca_engine_commit_packed without a preceding dca_engine_pop_expand_packed");
        return DCA_E_STATE;
    }
    DCA_ARG(h != nullptr || e->pk_rows == 0);
    hipStream_t s = (hipStream_t)stream;
    if (e->pk_rows > 0)
        DCA_HIP(hipMemcpyAsync(e->pk_h, h, (size_t)e->pk_rows * sizeof(float), hipMemcpyDeviceToDevice, s));
    e->phase = 0;
    return enqueue_commit(e, true, s);
}

// The next chunk of a run of iterations that starts at iteration `host_iter` of the search: up to the next rebase-period
// boundary first (so that a steady search replays whole periods, whose pattern repeats), then whole periods, kGraphChunk
// iterations at most; during the ramp up to its end.  mask: bit i = iteration i of the chunk is a rebase iteration.
static void plan_chunk(long host_iter, int remaining, bool single_only, int* n_out, unsigned long long* mask_out) {
    int n = remaining;
    if (single_only) {
        n = 1;
    } else if (host_iter < kRampIters) {
        n = n < (int)(kRampIters - host_iter) ? n : (int)(kRampIters - host_iter);
    } else {
        const int to_boundary = (int)(kRefillPeriod - host_iter % kRefillPeriod) % kRefillPeriod;
        if (to_boundary > 0 && n > to_boundary)
            n = to_boundary;
        else if (to_boundary == 0 && n > kRefillPeriod)
            n = (n < kGraphChunk ? n : kGraphChunk) / kRefillPeriod * kRefillPeriod;
    }
    unsigned long long mask = 0;
    for (int it = 0; it < n; it++)
        if (rebase_due(host_iter + it)) mask |= 1ull << it;
    *n_out = n;
    *mask_out = mask;
}

/* host-only (no device): how dca_engine_run_builtin cuts a run into hipGraph chunks — exposed for the CPU tests */
int dca_engine_plan_chunk(int64_t host_iter, int remaining, int* n, uint64_t* rebase_mask) {
    DCA_ARG(host_iter >= 0 && remaining >= 1 && n != nullptr && rebase_mask != nullptr);
    unsigned long long m = 0;
    plan_chunk((long)host_iter, remaining, false, n, &m);
    *rebase_mask = (uint64_t)m;
    return 0;
}

int dca_engine_run_builtin(dca_engine* e, int heur_id, int iters, int use_graph, void* stream) {
    DCA_ARG(e != nullptr && heur_id >= 0 && heur_id <= DCA_HEUR_MANHATTAN && iters >= 0);
    if (e->phase != 0) {
        set_error("dca_engine_run_builtin between pop_expand and commit");
        return DCA_E_STATE;
    }
    hipStream_t s = (hipStream_t)stream;
    if (!use_graph) {
        for (int i = 0; i < iters; i++) {
            if (int rc = enqueue_first_half(e, heur_id, rebase_due(e->host_iter++), s)) return rc;
            if (int rc = enqueue_second_half(e, s)) return rc;
        }
        return 0;
    }
    if (e->graph_heur != heur_id) {
        drop_graphs(e);
        e->graph_heur = heur_id;
    }
    for (int i = 0; i < iters;) {
        int n = 0;
        unsigned long long mask = 0;
        plan_chunk(e->host_iter, iters - i, h_tune[7] != 0, &n, &mask);  // (knob 7: single-iteration graphs only)
        dca_engine::GraphSlot* g = nullptr;
        for (int q = 0; q < kGraphSlots && !g; q++)
            if (e->gslot[q].exec && e->gslot[q].n == n && e->gslot[q].mask == mask) g = &e->gslot[q];
        if (!g) {
            for (int q = 0; q < kGraphSlots && !g; q++)
                if (!e->gslot[q].exec) g = &e->gslot[q];
            if (!g) {
                g = &e->gslot[e->gslot_next];
                e->gslot_next = (e->gslot_next + 1) % kGraphSlots;
                drop_graph_slot(*g);
            }
            hipStream_t cs;
            DCA_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
            hipError_t err = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
            int rc = 0;
            if (err == hipSuccess) {
                for (int it = 0; it < n && !rc; it++) {
                    rc = enqueue_first_half(e, heur_id, (mask >> it) & 1ull, cs);
                    if (!rc) rc = enqueue_second_half(e, cs);
                }
                err = hipStreamEndCapture(cs, &g->graph);
            }
            (void)hipStreamDestroy(cs);
            if (err != hipSuccess) {
                drop_graph_slot(*g);
                return hip_fail(err, "hipStream capture");
            }
            if (rc) {
                drop_graph_slot(*g);
                return rc;
            }
            if (hipError_t ierr = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0); ierr != hipSuccess) {
                drop_graph_slot(*g);
                return hip_fail(ierr, "hipGraphInstantiate");
            }
            g->n = n;
            g->mask = mask;
        }
        DCA_HIP(hipGraphLaunch(g->exec, s));
        e->host_iter += n;
        i += n;
    }
    return 0;
}

int dca_engine_profile_builtin(dca_engine* e, int heur_id, int iters, int use_graph, float* span_ms, float* gap_ms,
                               void* stream) {
    // Device-side profile of `iters` iterations of run_builtin, eager or as hipGraph replays: every workgroup stamps
    // the device wall clock at entry and exit (Stamp), per launch the host takes max(end) - min(start) as the launch's
    // busy span and min(start of the next launch) - max(end) as the gap in front of it.  Slots (span_ms / gap_ms
    // [DCA_PROF_SLOTS], summed over the iterations): 0-2 refill hist/scan/move (every 8th iteration), 3 sel_hist,
    // 4 sel_scan, 5 sel_collect, 6 rank, 7 expand, 8 probe, 9 decide, 10 pack (dedup-first stepping only), 11 commit.
    // gap_ms[k] = idle time in front of launch k inside an iteration (the first launch of an iteration has none).
    DCA_ARG(e != nullptr && span_ms != nullptr && heur_id >= 0 && heur_id <= DCA_HEUR_MANHATTAN && iters >= 0);
    if (e->phase != 0) {
        set_error("dca_engine_profile_builtin between pop_expand and commit");
        return DCA_E_STATE;
    }
    hipStream_t s = (hipStream_t)stream;
    constexpr size_t kWords = (size_t)P_COUNT * kProfSlots * 2;
    if (e->d_prof == nullptr) {
        if (int rc = dev_alloc(e, &e->d_prof, kWords)) return rc;
        DCA_HIP(hipHostMalloc((void**)&e->h_prof, kWords * sizeof(unsigned long long), hipHostMallocDefault));
    }
    int rate_khz = 100000;  // s_memrealtime: 100 MHz on gfx950
    (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    if (rate_khz <= 0) rate_khz = 100000;
    for (int k = 0; k < DCA_PROF_SLOTS; k++) {
        span_ms[k] = 0.f;
        if (gap_ms) gap_ms[k] = 0.f;
    }
    DCA_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < e->K; i++) e->E[i].prof = e->d_prof;
    if (int rc = upload_engs(e)) return rc;
    int rc = 0;
    for (int it = 0; it < iters && !rc; it++) {
        for (size_t w = 0; w < kWords; w += 2) {
            e->h_prof[w] = ~0ull;
            e->h_prof[w + 1] = 0ull;
        }
        if (hipMemcpyAsync(e->d_prof, e->h_prof, kWords * sizeof(unsigned long long), hipMemcpyHostToDevice, s) != hipSuccess)
            rc = hip_fail(hipGetLastError(), "hipMemcpyAsync(prof)");
        if (!rc) rc = dca_engine_run_builtin(e, heur_id, 1, use_graph, stream);
        if (!rc && hipMemcpyAsync(e->h_prof, e->d_prof, kWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, s) != hipSuccess)
            rc = hip_fail(hipGetLastError(), "hipMemcpyAsync(prof)");
        if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = hip_fail(hipGetLastError(), "hipStreamSynchronize");
        unsigned long long prev_end = 0;
        for (int k = 0; k < P_COUNT && !rc; k++) {
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int b = 0; b < kProfSlots; b++) {
                const unsigned long long a = e->h_prof[((size_t)k * kProfSlots + b) * 2], z = e->h_prof[((size_t)k * kProfSlots + b) * 2 + 1];
                if (a != ~0ull && a < t0) t0 = a;
                if (z > t1) t1 = z;
            }
            if (t0 == ~0ull || t1 < t0) continue;  // launch not part of this iteration (or it exited before stamping)
            span_ms[k] += (float)((double)(t1 - t0) / (double)rate_khz);
            if (gap_ms && prev_end != 0 && t0 > prev_end) gap_ms[k] += (float)((double)(t0 - prev_end) / (double)rate_khz);
            prev_end = t1;
        }
    }
    for (int i = 0; i < e->K; i++) e->E[i].prof = nullptr;
    if (int urc = upload_engs(e)) return urc;
    return rc;
}

int dca_debug_tune(int knob, int value) {
    if (knob >= 0 && knob < 16) h_tune[knob] = value;  // (host-side knobs: 4 = workgroups of k_sel_collect; set before the first step)
    // diagnostics: 0 extra log2 of sub-bins per large bin, 1 sub-bin size above which a sub-bin is refined on its own,
    // 2 BACK squeeze mark in 1/1024ths of max_nodes, 3 threshold-bin size above which the grid refines the bin (giant
    // iterations), 5 (host, before create) giant-bin path off, 6 (host) k_sel_scan launched in every iteration, 7 (host) single-iteration graphs only, 9 largest bin ranked a thread per entry
    DCA_ARG(knob >= 0 && knob < 16);
    DCA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_tune), &value, sizeof(int), (size_t)knob * sizeof(int), hipMemcpyHostToDevice));
    return 0;
}

int dca_engine_set_weight_instance(dca_engine* e, int inst, double weight) {
    DCA_INST(e, inst);
    DCA_ARG(weight >= 0.0);
    if (e->phase != 0) {
        set_error("dca_engine_set_weight_instance between pop_expand and commit");
        return DCA_E_STATE;
    }
    if (int rc = pull_dev_weights(e)) return rc;  // (the other instances may carry device-set weights)
    e->E[inst].w = weight;
    e->E[inst].wf = (float)weight;
    DCA_HIP(hipDeviceSynchronize());  // (no launch may still be reading the instance array)
    return upload_engs(e);
}

int dca_engine_set_weights(dca_engine* e, const double* weights, int n) {
    DCA_ARG(e != nullptr && weights != nullptr && n >= 1 && n <= e->K);
    if (e->phase != 0) {
        set_error("dca_engine_set_weights between pop_expand and commit");
        return DCA_E_STATE;
    }
    for (int i = 0; i < n; i++) DCA_ARG(weights[i] >= 0.0);
    if (int rc = pull_dev_weights(e)) return rc;  // (instances n..K-1 may carry device-set weights)
    for (int i = 0; i < n; i++) {
        e->E[i].w = weights[i];
        e->E[i].wf = (float)weights[i];
    }
    DCA_HIP(hipDeviceSynchronize());  // (no launch may still be reading the instance array)
    return upload_engs(e);
}

int dca_engine_park_instance(dca_engine* e, int inst, void* stream) {
    DCA_INST(e, inst);
    if (e->phase != 0) {  // its k_commit would return early and never record the slots the expansion half claimed
        set_error("dca_engine_park_instance between pop_expand and commit");
        return DCA_E_STATE;
    }
    const int32_t one = 1;
    DCA_HIP(hipStreamSynchronize((hipStream_t)stream));
    DCA_HIP(hipMemcpy(&e->E[inst].ctl->done, &one, sizeof(one), hipMemcpyHostToDevice));
    return 0;
}

/* The batched, synchronisation-free forms an ASTAR update steps thousands of batch-1 searches with (updater.py:36-54; ADVICE
 * r04: a host round trip per instance and per step made that update launch- and sync-bound). */
int dca_engine_reset_many(dca_engine* e, const uint8_t* roots_dev, int n, void* stream) {
    DCA_ARG(e != nullptr && roots_dev != nullptr && n >= 0 && n <= e->K);
    hipStream_t s = (hipStream_t)stream;
    if (e->phase != 0)  // (an abandoned half iteration: see dca_engine_reset_instance)
        for (int i = 0; i < e->K; i++) e->tab_cleared[i] = 0;
    for (int i = 0; i < e->K; i++) {
        Eng& E = e->E[i];
        if (i >= n) {  // parked: every launch of this instance is a no-op until it is reset again
            DCA_HIP(hipMemsetD32Async((hipDeviceptr_t)&E.ctl->done, 1, 1, s));
            continue;
        }
        DCA_HIP(hipMemcpyAsync(E.state, roots_dev + (size_t)i * (size_t)E.D, (size_t)E.D, hipMemcpyDeviceToDevice, s));
        const int force = e->tab_cleared[i] == 0 ? 1 : 0;
        // (both are enqueued, one of them returns at once — see clear_by_list; grids sized for the table: these are small searches)
        const unsigned g_all = (unsigned)std::min<uint64_t>(4096, ((uint64_t)E.tab_cap + 255) / 256);
        const unsigned g_list = (unsigned)std::min<uint64_t>(1024, ((uint64_t)E.tab_cap / 8 + 255) / 256);
        hipLaunchKernelGGL(k_init_table, dim3(g_all), dim3(256), 0, s, E.tab, E.tab_cap, E.ctl, force);
        hipLaunchKernelGGL(k_clear_table_list, dim3(g_list ? g_list : 1), dim3(256), 0, s, E.tab, E.tab_cap, E.closed_slots, E.ctl, force);
        e->tab_cleared[i] = 1;
        hipLaunchKernelGGL(k_reset, dim3(1), dim3(64), 0, s, E);
        DCA_HIP(hipMemsetAsync(E.child_multi, 0, (size_t)E.M, s));
    }
    e->phase = 0;
    e->host_iter = 0;
    return launch_check("k_reset (many)");
}

int dca_engine_root_commit_many(dca_engine* e, const float* h_roots_dev, int n, void* stream) {
    DCA_ARG(e != nullptr && n >= 0 && n <= e->K);
    if (e->E[0].sem != DCA_SEM_PY) return 0;
    DCA_ARG(h_roots_dev != nullptr || n == 0);
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_root_commit, dim3(1), dim3(64), 0, (hipStream_t)stream, e->E[i], h_roots_dev + i);
    return launch_check("k_root_commit (many)");
}

int dca_engine_set_weights_dev(dca_engine* e, const double* weights_dev, int n, void* stream) {
    DCA_ARG(e != nullptr && weights_dev != nullptr && n >= 1 && n <= e->K);
    if (e->phase != 0) {
        set_error("dca_engine_set_weights_dev between pop_expand and commit");
        return DCA_E_STATE;
    }
    hipLaunchKernelGGL(k_set_weights, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, e->d_engs, weights_dev, n);
    e->w_dev = true;
    return launch_check("k_set_weights");
}

int dca_engine_last_popped(dca_engine* e, uint8_t* states, uint8_t* flags, void* stream) {
    DCA_ARG(e != nullptr && states != nullptr && flags != nullptr);
    if (e->phase == 0) {
        set_error("dca_engine_last_popped outside an iteration (call it between pop_expand and commit)");
        return DCA_E_STATE;
    }
    const Eng& E = e->E[0];
    const unsigned gx = (unsigned)(E.B < 1024 ? E.B : 1024);
    hipLaunchKernelGGL(k_gather_popped, gxy(gx, e), dim3(64), 0, (hipStream_t)stream, e->d_engs, states, flags);
    return launch_check("k_gather_popped");
}

int dca_engine_info(dca_engine* e, int64_t* out, void* stream) {
    DCA_ARG(e != nullptr && out != nullptr);
    uint32_t off = 0;
    DCA_HIP(hipStreamSynchronize((hipStream_t)stream));
    DCA_HIP(hipMemcpy(&off, e->E[0].coop_off, sizeof(off), hipMemcpyDeviceToHost));
    out[0] = (int64_t)e->collect_blocks;
    out[1] = (int64_t)e->collect_resident;
    out[2] = (int64_t)e->E[0].coop;
    out[3] = (int64_t)e->E[0].tab_cap * (int64_t)sizeof(Slot);
    out[4] = (int64_t)off;
    out[5] = out[6] = out[7] = 0;
    return 0;
}

int dca_engine_set_tiers(dca_engine* e, int64_t front_keep, int64_t front_max) {
    // test / tuning hook: FRONT hysteresis in entries (defaults 32*B and 96*B).  Search results never depend on it.
    DCA_ARG(e != nullptr && front_keep >= 1 && front_max >= front_keep && front_max < (1ll << 31));
    for (int i = 0; i < e->K; i++) {
        e->E[i].f_keep = (uint32_t)front_keep;
        e->E[i].f_max = (uint32_t)front_max;
    }
    drop_graphs(e);  // (kernel arguments are only the instance-array pointer, but stay on the safe side)
    return upload_engs(e);
}

int dca_engine_status_instance(dca_engine* e, int inst, dca_status* out, void* stream) {
    DCA_INST(e, inst);
    DCA_ARG(out != nullptr);
    if (int rc = fetch_ctl(e, inst, (hipStream_t)stream)) return rc;
    const Ctl& c = *e->h_ctl;
    // between the two halves of an iteration the expansion's record (S[(iters + 1) & 1]) is not current yet
    const IterState& S = c.S[(c.iters + (e->phase != 0 ? 1 : 0)) & 1];
    out->done = c.done;
    out->failed = c.failed;
    out->iterations = c.iters;
    out->nodes_generated = c.gen;
    out->nodes_expanded = c.expanded;
    out->open_size = (int64_t)c.open_n[c.cur_f].v - (int64_t)c.front_dead.v + (int64_t)c.open_n[c.cur_b].v - (int64_t)c.back_dead.v;
    out->closed_size = c.closed_n.v;
    out->pool_size = S.pool_n;
    out->best_cost = S.has_best ? (double)S.best_cost : __builtin_nan("");
    return 0;
}
int dca_engine_status(dca_engine* e, dca_status* out, void* stream) {
    return dca_engine_status_instance(e, 0, out, stream);
}

int dca_engine_debug(dca_engine* e, double* out, void* stream) {
    DCA_ARG(e != nullptr && out != nullptr);
    if (int rc = fetch_ctl(e, 0, (hipStream_t)stream)) return rc;
    const Ctl& c = *e->h_ctl;
    auto cost = [](uint64_t k) {
        uint64_t b = (k & 0x8000000000000000ull) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
        double d;
        memcpy(&d, &b, 8);
        return d;
    };
    const IterState& S = c.S[(c.iters + (e->phase != 0 ? 1 : 0)) & 1];
    out[0] = (double)c.open_n[c.cur_f].v - (double)c.front_dead.v;
    out[1] = (double)c.open_n[c.cur_b].v - (double)c.back_dead.v;
    out[2] = cost(c.rng[c.cur_f].kmin);
    out[3] = cost(c.rng[c.cur_f].kmax);
    out[4] = cost(c.rng[c.cur_b].kmin);
    out[5] = cost(c.rng[c.cur_b].kmax);
    out[6] = cost(c.T);
    out[7] = c.want;
    out[8] = c.bstar;
    out[9] = c.dbg_nord;
    out[10] = c.dbg_maxbin;
    out[11] = c.dbg_giant_seen;
    out[12] = c.dbg_maxsub;
    out[13] = c.spill_bin;
    out[14] = S.npop;
    out[15] = S.m;
    return 0;
}

int dca_engine_last_children(dca_engine* e, const uint8_t** states, int64_t* m_live, void* stream) {
    DCA_ARG(e != nullptr && states != nullptr && m_live != nullptr);
    if (int rc = fetch_ctl(e, 0, (hipStream_t)stream)) return rc;
    const IterState& S = e->h_ctl->S[(e->h_ctl->iters + (e->phase != 0 ? 1 : 0)) & 1];
    *states = e->E[0].state + (size_t)S.base * e->E[0].D;
    *m_live = S.m;
    return 0;
}

int dca_engine_solution_instance(dca_engine* e, int inst, int32_t* moves, int cap, int* len, double* path_cost,
                                 void* stream) {
    DCA_INST(e, inst);
    DCA_ARG(len != nullptr && cap >= 0 && (cap == 0 || moves != nullptr));
    hipStream_t s = (hipStream_t)stream;
    const Eng& E = e->E[inst];
    hipLaunchKernelGGL(k_solution, dim3(1), dim3(64), 0, s, E, E.d_moves, e->d_cost);
    if (int rc = launch_check("k_solution")) return rc;
    DCA_HIP(hipMemcpyAsync(e->h_moves, E.d_moves, kMaxMoves * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    DCA_HIP(hipMemcpyAsync(e->h_cost, e->d_cost, sizeof(double), hipMemcpyDeviceToHost, s));
    DCA_HIP(hipStreamSynchronize(s));
    int n = e->h_moves[0];
    if (n < 0) {
        set_error("no solution recorded (search not finished)");
        return DCA_E_NOTFOUND;
    }
    *len = n;
    if (path_cost) *path_cost = *e->h_cost;
    for (int i = 0; i < n && i < cap; i++) moves[i] = e->h_moves[1 + i];
    return 0;
}
int dca_engine_solution(dca_engine* e, int32_t* moves, int cap, int* len, double* path_cost, void* stream) {
    return dca_engine_solution_instance(e, 0, moves, cap, len, path_cost, stream);
}

}  // extern "C"