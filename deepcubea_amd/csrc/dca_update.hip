// dca_update.hip — device pieces of the approximate-value-iteration UPDATE STEP (BASELINE
// configs[4]): training-state generation by random reverse walks and the Bellman backup over expanded children.
//
// Reference behaviour restated (paths relative to forestagostinelli/DeepCubeA):
//   environments/cube3.py:96-127, n_puzzle.py:100-134   generate_states: k_i ~ U{lo..hi} reverse moves from the goal
//   utils/search_utils.py:16-32                         bellman: ctg_backup = min_a(tc + h(child_a)) * !is_solved(state)
//   search_methods/gbfs.py:86-120                       greedy move = argmin_a(tc + h(child_a)) (first index on ties)
// The expansion itself is dca_*_expand_fused (dca_env.hip); the heuristic runs on PyTorch-ROCm in between.
#include "dca_common.h"
#include "dca_tile.h"

namespace dca {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// One lane per state; the state lives in LDS (two rows per lane, ping-pong).  Every lane draws its own walk length
// and its own uniformly random move per step from a counter-based generator keyed by (seed, state index, step):
// the per-state marginal of the reference's procedure (which moves random subsets together) is this same walk.
template <int ENV, int DIM>
__global__ __launch_bounds__(64) void scramble_kernel(int64_t n, int lo, int hi, uint64_t seed, int64_t index0,
                                                      uint8_t* __restrict__ out, int32_t* __restrict__ out_k,
                                                      int8_t* __restrict__ out_moves, int moves_stride) {
    using E = EnvT<ENV, DIM>;
    __shared__ uint8_t rows[2][64][E::D + 2];
    __shared__ uint8_t perm[ENV == DCA_ENV_CUBE3 ? 12 * 54 : 4];
    const int lane = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    if constexpr (ENV == DCA_ENV_CUBE3)
        for (int t = lane; t < 12 * 54; t += 64) perm[t] = d_cube3_perm.p[t / 54][t % 54];
    __syncthreads();
    if (i >= n) return;
    uint8_t* cur = rows[0][lane];
    uint8_t* nxt = rows[1][lane];
    for (int j = 0; j < E::D; j++) cur[j] = (uint8_t)goal_byte(ENV, E::D, j);  // goal
    int z = E::D - 1;                                                                           // puzzle blank
    const uint64_t key = splitmix64(seed ^ splitmix64((uint64_t)(index0 + i)));
    const uint32_t range = (uint32_t)(hi - lo + 1);
    const int k = lo + (int)(((splitmix64(key) >> 32) * (uint64_t)range) >> 32);
    for (int t = 0; t < k; t++) {
        const uint32_t r = (uint32_t)(splitmix64(key + 0x632BE59BD9B4E019ull * (uint64_t)(t + 1)) >> 32);
        const int a = (int)(((uint64_t)r * (uint64_t)E::A) >> 32);  // the move taken in reverse
        const int ra = a ^ 1;                                        // prev_state(a) = next_state(a^1)
        if constexpr (ENV == DCA_ENV_CUBE3) {
            for (int j = 0; j < 54; j++) nxt[j] = cur[perm[ra * 54 + j]];
            uint8_t* tmp = cur;
            cur = nxt;
            nxt = tmp;
        } else if constexpr (ENV == DCA_ENV_LIGHTSOUT) {
            // every move is its own inverse (lights_out.py:52-53): the walk presses cell a
            for (int j = 0; j < E::D; j++)
                if (lightsout_flip(DIM, a, j)) cur[j] = (uint8_t)((cur[j] + 1) & 1);
            (void)ra;
        } else {
            const int s = npuzzle_swap(DIM, z, ra);
            cur[z] = cur[s];
            cur[s] = 0;
            z = s;
        }
        if (out_moves) out_moves[i * moves_stride + t] = (int8_t)a;
    }
    for (int j = 0; j < E::D; j++) out[i * E::D + j] = cur[j];
    if (out_k) out_k[i] = k;
}

// ctg_backup[i] = solved[i] ? 0 : min_a(1 + max(h[i*A+a], 0));  argmin = first index of the minimum
__global__ void bellman_backup_kernel(const float* __restrict__ h, const uint8_t* __restrict__ solved, int64_t n, int A,
                                      int clip_zero, float* __restrict__ ctg, int32_t* __restrict__ argmin) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float best = 0.f;
    int bi = 0;
    for (int a = 0; a < A; a++) {
        float v = h[i * A + a];
        if (clip_zero) v = fmaxf(v, 0.f);
        v = __fadd_rn(1.0f, v);  // transition cost 1.0 + cost-to-go of the child
        if (a == 0 || v < best) {
            best = v;
            bi = a;
        }
    }
    if (ctg) ctg[i] = (solved && solved[i]) ? 0.f : best;
    if (argmin) argmin[i] = bi;
}

template <int ENV, int DIM>
static int launch_scramble(int64_t n, int lo, int hi, uint64_t seed, int64_t index0, uint8_t* out, int32_t* out_k,
                           int8_t* out_moves, int stride, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL((scramble_kernel<ENV, DIM>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, n, lo, hi, seed,
                       index0, out, out_k, out_moves, stride);
    return launch_check("scramble_kernel");
}

}  // namespace dca

using namespace dca;

extern "C" {

int dca_generate_states(int env, int dim, int64_t n, int back_lo, int back_hi, uint64_t seed, int64_t index0,
                        uint8_t* out_states, int32_t* out_num_back, int8_t* out_moves, int moves_stride,
                        void* stream) {
    DCA_ARG(n >= 0 && back_lo >= 0 && back_hi >= back_lo && back_hi < (1 << 20));
    DCA_ARG(n == 0 || out_states != nullptr);
    DCA_ARG(out_moves == nullptr || moves_stride >= back_hi);
    DCA_ARG((n + 63) / 64 < (1ll << 31));
    hipStream_t s = (hipStream_t)stream;
    if (env == DCA_ENV_CUBE3)
        return launch_scramble<DCA_ENV_CUBE3, 0>(n, back_lo, back_hi, seed, index0, out_states, out_num_back, out_moves,
                                                 moves_stride, s);
    if (env == DCA_ENV_LIGHTSOUT && dim == 7)
        return launch_scramble<DCA_ENV_LIGHTSOUT, 7>(n, back_lo, back_hi, seed, index0, out_states, out_num_back, out_moves,
                                                     moves_stride, s);
    switch (env == DCA_ENV_NPUZZLE ? dim : -1) {
        case 4: return launch_scramble<DCA_ENV_NPUZZLE, 4>(n, back_lo, back_hi, seed, index0, out_states, out_num_back, out_moves, moves_stride, s);
        case 5: return launch_scramble<DCA_ENV_NPUZZLE, 5>(n, back_lo, back_hi, seed, index0, out_states, out_num_back, out_moves, moves_stride, s);
        case 6: return launch_scramble<DCA_ENV_NPUZZLE, 6>(n, back_lo, back_hi, seed, index0, out_states, out_num_back, out_moves, moves_stride, s);
        case 7: return launch_scramble<DCA_ENV_NPUZZLE, 7>(n, back_lo, back_hi, seed, index0, out_states, out_num_back, out_moves, moves_stride, s);
    }
    set_error("unknown env %d / dim %d", env, dim);
    return DCA_E_BADARG;
}

int dca_bellman_backup(const float* h_children, const uint8_t* solved_parent, int64_t n, int num_moves, int clip_zero,
                       float* ctg_backup, int32_t* argmin, void* stream) {
    DCA_ARG(n >= 0 && num_moves > 0 && num_moves <= 256 && (n == 0 || h_children != nullptr));
    if (n == 0) return 0;
    hipLaunchKernelGGL(bellman_backup_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       h_children, solved_parent, n, num_moves, clip_zero, ctg_backup, argmin);
    return launch_check("bellman_backup_kernel");
}

}  // extern "C"
