/*
 * dca_oracle.cpp — CPU restatement of the reference's BWAS hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load liboracle.so; the product (deepcubea_amd/, libdca_hip.so)
 * never links, loads or calls anything in this directory.
 *
 * Every function cites the reference lines it restates (paths relative to
 * forestagostinelli/DeepCubeA).  Pinned by tests/test_oracle_golden.py against
 *   - tests/golden/golden.npz  (outputs of the reference's Python implementation imported
 *     in the build container: env ops, python-semantics A* traces), and
 *   - the known-answer rows of SURVEY.md Appendix A for the cpp semantics (produced by the
 *     reference binary during the survey; the binary itself is NOT buildable here because
 *     parallel_weighted_astar.cpp:30 needs boost, which this image lacks — so the cpp
 *     semantics are pinned by those recorded outputs only: "parity partially pinned").
 *
 * Build: make -C oracle   (g++ -O3 -fopenmp -shared)
 */
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <queue>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ------------------------------------------------------------------------------------------
// cube3 move tables as 4-cycles of the six "-1" turns (sticker = face*9+row*3+col, faces
// U D L R B F).  (a b c d): next[a]=cur[b], next[b]=cur[c], next[c]=cur[d], next[d]=cur[a].
// Same permutation as Cube3._compute_rotation_idxs (environments/cube3.py:183-256) and the
// literal rotateIdxs_old/new tables (cpp/environments.h:75-105).
// ------------------------------------------------------------------------------------------
const int kCycles[6][5][4] = {
    {{0, 2, 8, 6}, {1, 5, 7, 3}, {20, 38, 29, 47}, {23, 41, 32, 50}, {26, 44, 35, 53}},
    {{9, 11, 17, 15}, {10, 14, 16, 12}, {18, 45, 27, 36}, {21, 48, 30, 39}, {24, 51, 33, 42}},
    {{0, 45, 9, 44}, {1, 46, 10, 43}, {2, 47, 11, 42}, {18, 20, 26, 24}, {19, 23, 25, 21}},
    {{6, 38, 15, 51}, {7, 37, 16, 52}, {8, 36, 17, 53}, {27, 29, 35, 33}, {28, 32, 34, 30}},
    {{2, 18, 15, 35}, {5, 19, 12, 34}, {8, 20, 9, 33}, {36, 38, 44, 42}, {37, 41, 43, 39}},
    {{0, 29, 17, 24}, {3, 28, 14, 25}, {6, 27, 11, 26}, {45, 47, 53, 51}, {46, 50, 52, 48}},
};

// ------------------------------------------------------------------------------------------
// cube4 (cpp/environments.cpp:263-370): the 4-cycles of the twelve "_n1" moves — six outer-layer turns (8 cycles: four inside
// the face's own 4x4 grid, four strips around it), then six inner-slice turns (4 cycles; unused rows are zero) — in the
// order of the reference's table list (U0 D0 L0 R0 B0 F0 U1 D1 L1 R1 B1 F1; action = 2 * index, + 1 for the "_1" inverse).
// (a b c d): next[a] = cur[b], next[b] = cur[c], next[c] = cur[d], next[d] = cur[a] — the same permutations as the reference's
// rotateIdxs_old / rotateIdxs_new pair lists (which repeat corner stickers; the repeats write the same value twice).
// Checked move by move against the reference's compiled class (oracle/_ref) in tests/test_cube4_cpu.py.
// ------------------------------------------------------------------------------------------
const int kCube4Cycles[12][8][4] = {
    {{0, 3, 15, 12}, {1, 7, 14, 8}, {2, 11, 13, 4}, {5, 6, 10, 9}, {35, 67, 51, 83}, {39, 71, 55, 87}, {43, 75, 59, 91}, {47, 79, 63, 95}},
    {{16, 19, 31, 28}, {17, 23, 30, 24}, {18, 27, 29, 20}, {21, 22, 26, 25}, {32, 80, 48, 64}, {36, 84, 52, 68}, {40, 88, 56, 72}, {44, 92, 60, 76}},
    {{0, 80, 16, 79}, {1, 81, 17, 78}, {2, 82, 18, 77}, {3, 83, 19, 76}, {32, 35, 47, 44}, {33, 39, 46, 40}, {34, 43, 45, 36}, {37, 38, 42, 41}},
    {{12, 67, 28, 92}, {13, 66, 29, 93}, {14, 65, 30, 94}, {15, 64, 31, 95}, {48, 51, 63, 60}, {49, 55, 62, 56}, {50, 59, 61, 52}, {53, 54, 58, 57}},
    {{3, 32, 28, 63}, {7, 33, 24, 62}, {11, 34, 20, 61}, {15, 35, 16, 60}, {64, 67, 79, 76}, {65, 71, 78, 72}, {66, 75, 77, 68}, {69, 70, 74, 73}},
    {{0, 51, 31, 44}, {4, 50, 27, 45}, {8, 49, 23, 46}, {12, 48, 19, 47}, {80, 83, 95, 92}, {81, 87, 94, 88}, {82, 91, 93, 84}, {85, 86, 90, 89}},
    {{34, 66, 50, 82}, {38, 70, 54, 86}, {42, 74, 58, 90}, {46, 78, 62, 94}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}},
    {{33, 81, 49, 65}, {37, 85, 53, 69}, {41, 89, 57, 73}, {45, 93, 61, 77}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}},
    {{4, 84, 20, 75}, {5, 85, 21, 74}, {6, 86, 22, 73}, {7, 87, 23, 72}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}},
    {{8, 71, 24, 88}, {9, 70, 25, 89}, {10, 69, 26, 90}, {11, 68, 27, 91}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}},
    {{2, 36, 29, 59}, {6, 37, 25, 58}, {10, 38, 21, 57}, {14, 39, 17, 56}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}},
    {{1, 55, 30, 40}, {5, 54, 26, 41}, {9, 53, 22, 42}, {13, 52, 18, 43}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}},
};

struct Tables {
    // scatter form used by the reference: newState[new_idx[a][i]] = state[old_idx[a][i]]
    // (cube3.py:167, environments.cpp:225-229); 20 distinct pairs per move.
    uint8_t new_idx[12][20];
    uint8_t old_idx[12][20];
    uint8_t perm[12][54];
    uint8_t swap[8][49][4];  // [dim][z][move] (n_puzzle.py:174-214, environments.cpp:4-46)
    Tables() {
        for (int a = 0; a < 12; a++)
            for (int i = 0; i < 54; i++) perm[a][i] = (uint8_t)i;
        for (int f = 0; f < 6; f++) {
            int k = 0;
            for (int c = 0; c < 5; c++)
                for (int j = 0; j < 4; j++, k++) {
                    int a = kCycles[f][c][j], b = kCycles[f][c][(j + 1) % 4];
                    new_idx[2 * f][k] = (uint8_t)a;      // "-1" turn = action 2f (cube3.py:28)
                    old_idx[2 * f][k] = (uint8_t)b;
                    new_idx[2 * f + 1][k] = (uint8_t)b;  // "+1" turn = inverse
                    old_idx[2 * f + 1][k] = (uint8_t)a;
                    perm[2 * f][a] = (uint8_t)b;
                    perm[2 * f + 1][b] = (uint8_t)a;
                }
        }
        for (int n = 4; n <= 7; n++)
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++) {
                    int z = i * n + j;
                    swap[n][z][0] = (uint8_t)(i < n - 1 ? z + n : z);  // U
                    swap[n][z][1] = (uint8_t)(i > 0 ? z - n : z);      // D
                    swap[n][z][2] = (uint8_t)(j < n - 1 ? z + 1 : z);  // L
                    swap[n][z][3] = (uint8_t)(j > 0 ? z - 1 : z);      // R
                }
    }
};
const Tables T;

enum { ENV_CUBE3 = 0, ENV_NPUZZLE = 1, ENV_LIGHTSOUT = 2, ENV_CUBE4 = 3 };
enum { SEM_PY = 0, SEM_CPP = 1 };

struct Env {
    int id, dim, D, A;
};
inline Env make_env(int id, int dim) {
    Env e;
    e.id = id;
    e.dim = dim;
    e.D = id == ENV_CUBE3 ? 54 : id == ENV_CUBE4 ? 96 : dim * dim;
    e.A = id == ENV_CUBE3 ? 12 : id == ENV_CUBE4 ? 24 : id == ENV_LIGHTSOUT ? dim * dim : 4;
    return e;
}

// Cube3::getNextState (environments.cpp:222-234) / Cube3._move_np (cube3.py:163-171)
inline void cube3_move(const uint8_t* s, int a, uint8_t* out) {
    memcpy(out, s, 54);
    for (int i = 0; i < 20; i++) out[T.new_idx[a][i]] = s[T.old_idx[a][i]];
}
// PuzzleN::getNextState (environments.cpp:92-104) / NPuzzle._move_np (n_puzzle.py:216-231);
// blank located by scan like n_puzzle.py:51-53
inline void npuzzle_move(const uint8_t* s, int dim, int a, uint8_t* out) {
    int D = dim * dim, z = 0;
    for (int i = 0; i < D; i++)
        if (s[i] == 0) {
            z = i;
            break;
        }
    memcpy(out, s, (size_t)D);
    int sw = T.swap[dim][z][a];
    out[z] = s[sw];
    out[sw] = 0;
}
// LightsOut::getNextState (environments.cpp:168-180) / LightsOut._move_np (lights_out.py:155-166): the five entries of the
// move matrix row (environments.cpp:133-154 / lights_out.py:33-44: the cell, right, left, up, down — an off-board
// neighbour repeats the cell itself) are each set to (old value + 1) % 2, always READ from the old state, so a repeated
// index is flipped once
inline void lightsout_move(const uint8_t* s, int dim, int a, uint8_t* out) {
    const int D = dim * dim;
    memcpy(out, s, (size_t)D);
    const int x = a / dim, y = a % dim;
    const int idx[5] = {a, x < dim - 1 ? a + dim : a, x > 0 ? a - dim : a, y < dim - 1 ? a + 1 : a, y > 0 ? a - 1 : a};
    for (int i = 0; i < 5; i++) out[idx[i]] = (uint8_t)((s[idx[i]] + 1) % 2);
}
// Cube4::getNextState (environments.cpp:329-343): newState[newIdx] = state[oldIdx] over the move's pair list — every source is
// read from the OLD state; here the pairs are the move's 4-cycles (action 2 m: next[c_j] = cur[c_{j+1}]; 2 m + 1: the inverse)
inline void cube4_move(const uint8_t* s, int a, uint8_t* out) {
    memcpy(out, s, 96);
    const int m = a >> 1;
    for (int c = 0; c < 8; c++) {
        const int* cy = kCube4Cycles[m][c];
        if (cy[0] == cy[1]) continue;  // unused row of an inner-slice move
        for (int j = 0; j < 4; j++) {
            const int x = cy[j], y = cy[(j + 1) % 4];
            if ((a & 1) == 0)
                out[x] = s[y];
            else
                out[y] = s[x];
        }
    }
}
inline void env_move(const Env& e, const uint8_t* s, int a, uint8_t* out) {
    if (e.id == ENV_CUBE4)
        cube4_move(s, a, out);
    else if (e.id == ENV_CUBE3)
        cube3_move(s, a, out);
    else if (e.id == ENV_LIGHTSOUT)
        lightsout_move(s, e.dim, a, out);
    else
        npuzzle_move(s, e.dim, a, out);
}
// Cube3::isSolved (environments.cpp:249-256), PuzzleN::isSolved (119-126)
inline bool env_solved(const Env& e, const uint8_t* s) {
    bool ok = true;
    if (e.id == ENV_CUBE4) {  // Cube4::isSolved (environments.cpp:355-365): every face shows one colour (sticker / 16)
        for (int side = 0; side < 6; side++)
            for (int i = 1; i < 16; i++) ok &= (s[side * 16 + i] / 16 == s[side * 16] / 16);
    } else if (e.id == ENV_CUBE3) {
        for (int i = 0; i < 54; i++) ok &= (s[i] == i);
    } else if (e.id == ENV_LIGHTSOUT) {  // LightsOut::isSolved (environments.cpp:196-204) / lights_out.py:65-68
        for (int i = 0; i < e.D; i++) ok &= (s[i] == 0);
    } else {
        for (int i = 0; i < e.D; i++) ok &= (s[i] == (uint8_t)((i + 1) % e.D));
    }
    return ok;
}

// include/dca.h state hash
inline uint64_t hash64(const uint8_t* s, int D) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ ((uint64_t)D * 0xD6E8FEB86659FD93ull);
    for (int k = 0; k < D; k += 8) {
        uint64_t w = 0;
        int nb = std::min(8, D - k);
        for (int j = 0; j < nb; j++) w |= (uint64_t)s[k + j] << (8 * j);
        h ^= w;
        h *= 0xFF51AFD7ED558CCDull;
        h ^= h >> 32;
    }
    h ^= h >> 33;
    h *= 0xC4CEB9FE1A85EC53ull;
    h ^= h >> 33;
    return h;
}

// include/dca.h DCA_HEUR_*
inline float heur_builtin(int id, const uint8_t* s, int D) {
    uint64_t sum = 0;
    for (int i = 0; i < D; i++) sum += (uint64_t)s[i] * (uint64_t)(7 * i + 3);
    switch (id) {
        case 0:
            return (float)(sum % 97) / 50.0f;
        case 1: {
            uint64_t x = (sum * 2654435761ull) & 0xFFFFFFFFull;
            return (float)((double)x / 4294967296.0 * 3.0);
        }
        case 2:
            return (float)(10.0 + 5.0 * ((double)(hash64(s, D) >> 11) / 9007199254740992.0));
        case 4: {  // DCA_HEUR_MANHATTAN (puzzles; 0 for cube3)
            int dim = D == 16 ? 4 : D == 25 ? 5 : D == 36 ? 6 : D == 49 ? 7 : 0, m = 0;
            if (!dim) return 0.0f;
            for (int p = 0; p < D; p++) {
                int t = s[p];
                if (t == 0) continue;
                int g = t - 1;
                m += std::abs(p / dim - g / dim) + std::abs(p % dim - g % dim);
            }
            return (float)m;
        }
        default:
            return 0.0f;
    }
}

typedef void (*heur_cb)(const uint8_t* states, int64_t n, int D, float* out, void* user);

struct HeurSrc {
    int builtin;  // >=0: DCA_HEUR_*, <0: callback
    heur_cb cb;
    void* user;
    void eval(const uint8_t* st, int64_t n, int D, float* out) const {
        if (builtin >= 0) {
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < n; i++) out[i] = heur_builtin(builtin, st + i * D, D);
        } else {
            cb(st, n, D, out, user);
        }
    }
};

struct StateKey {
    const uint8_t* p;
    int D;
};
struct KeyHash {
    size_t operator()(const StateKey& k) const { return (size_t)hash64(k.p, k.D); }
};
struct KeyEq {
    bool operator()(const StateKey& a, const StateKey& b) const { return memcmp(a.p, b.p, (size_t)a.D) == 0; }
};

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" {

struct oracle_result {
    int32_t solved;
    int32_t num_moves;
    double path_cost;
    int64_t nodes_generated;
    int64_t iterations;
    int64_t nodes_expanded;
    double seconds;       // wall time of the search loop
    int64_t open_size;    // at exit
    int64_t closed_size;  // at exit
};

const uint8_t* oracle_cube3_perm_table(void) { return &T.perm[0][0]; }
void oracle_npuzzle_swap_table(int dim, uint8_t* out) { memcpy(out, T.swap[dim], (size_t)dim * dim * 4); }

void oracle_next_state(int env, int dim, const uint8_t* in, int64_t n, int action, uint8_t* out) {
    Env e = make_env(env, dim);
    for (int64_t i = 0; i < n; i++) env_move(e, in + i * e.D, action, out + i * e.D);
}

// Cube3.expand (cube3.py:129-161) / getNextStates (environments.cpp:236-243): children [n,A,D]
void oracle_expand(int env, int dim, const uint8_t* in, int64_t n, uint8_t* children, uint8_t* solved,
                   uint64_t* hash, int threads) {
    Env e = make_env(env, dim);
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++)
        for (int a = 0; a < e.A; a++) {
            uint8_t* c = children + (i * e.A + a) * e.D;
            env_move(e, in + i * e.D, a, c);
            if (solved) solved[i * e.A + a] = env_solved(e, c);
            if (hash) hash[i * e.A + a] = hash64(c, e.D);
        }
}

void oracle_is_solved(int env, int dim, const uint8_t* in, int64_t n, uint8_t* out) {
    Env e = make_env(env, dim);
    for (int64_t i = 0; i < n; i++) out[i] = env_solved(e, in + i * e.D);
}
void oracle_hash64(const uint8_t* in, int64_t n, int D, uint64_t* out) {
    for (int64_t i = 0; i < n; i++) out[i] = hash64(in + i * D, D);
}
void oracle_heur_builtin(int id, const uint8_t* in, int64_t n, int D, float* out) {
    for (int64_t i = 0; i < n; i++) out[i] = heur_builtin(id, in + i * D, D);
}
// state_to_nnet_input (cube3.py:77-85: //9 ; n_puzzle.py:84-89: identity)
void oracle_nnet_input(int env, const uint8_t* in, int64_t n, int D, uint8_t* out) {
    for (int64_t i = 0; i < n * D; i++) out[i] = env == ENV_CUBE3 ? (uint8_t)(in[i] / 9) : in[i];
}
// F.one_hot(...).float().view(-1, D*depth) (pytorch_models.py:49-52)
void oracle_onehot_f32(const uint8_t* idx, int64_t n, int D, int depth, float* out) {
    memset(out, 0, (size_t)n * D * depth * sizeof(float));
    for (int64_t i = 0; i < n; i++)
        for (int j = 0; j < D; j++) out[(i * D + j) * depth + idx[i * D + j]] = 1.0f;
}

/* ==========================================================================================
 * BWAS, python semantics — search_methods/astar.py:18-340 (Node, Instance, AStar) for ONE
 * instance, restated over arrays.  Costs are float64, OPEN orders by
 * (cost, push count) (astar.py:64-67), CLOSED maps state -> best path cost and starts empty
 * (astar.py:55), children are deduplicated sequentially in (pop order, move) order
 * (astar.py:78-90), the search stops as soon as a popped node is solved (astar.py:421,73).
 * trace (optional) gets (|OPEN|, |CLOSED|, generated) after each iteration.
 * ========================================================================================== */
static int astar_py(const Env& e, const uint8_t* root, const HeurSrc& H, double w, int B, int64_t max_iters,
                    oracle_result* res, int32_t* moves_out, int moves_cap, int64_t* trace, int64_t trace_cap,
                    int stop_on_goal) {
    const int D = e.D, A = e.A;
    struct N {
        double g, cost;
        int64_t parent;
        int8_t move;
        bool solved;
    };
    std::vector<uint8_t> pool(root, root + D);
    std::vector<N> nodes;
    struct Ent {
        double cost;
        int64_t cnt;
        int64_t id;
    };
    auto cmp = [](const Ent& a, const Ent& b) { return a.cost > b.cost || (a.cost == b.cost && a.cnt > b.cnt); };
    std::priority_queue<Ent, std::vector<Ent>, decltype(cmp)> open(cmp);
    std::unordered_map<uint64_t, std::vector<std::pair<int64_t, double>>> closed_buckets;  // hash -> (rep node, g)
    int64_t closed_size = 0;

    float h0;
    H.eval(root, 1, D, &h0);
    bool rs = env_solved(e, root);
    double hr = std::max((double)h0, 0.0);
    // astar.py:196: weights*path_costs + heuristics*logical_not(is_solved), float64
    double c0 = w * 0.0 + hr * (rs ? 0.0 : 1.0);
    nodes.push_back({0.0, c0, -1, -1, rs});
    int64_t cnt = 0;
    open.push({c0, cnt++, 0});

    int64_t gen = 0, it = 0, expanded = 0;
    std::vector<int64_t> goals, popped;
    std::vector<uint8_t> ch;
    std::vector<float> hv;
    std::vector<uint8_t> sv;
    double t0 = now_s();
    while ((goals.empty() || !stop_on_goal) && it < max_iters && !open.empty()) {
        int64_t npop = std::min<int64_t>(B, (int64_t)open.size());
        popped.clear();
        for (int64_t i = 0; i < npop; i++) {
            popped.push_back(open.top().id);
            open.pop();
        }
        for (int64_t id : popped)
            if (nodes[id].solved) goals.push_back(id);
        int64_t m = npop * A;
        ch.resize((size_t)m * D);
        sv.resize((size_t)m);
        hv.resize((size_t)m);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < npop; i++)
            for (int a = 0; a < A; a++) {
                uint8_t* c = &ch[(size_t)(i * A + a) * D];
                env_move(e, &pool[(size_t)popped[i] * D], a, c);
                sv[i * A + a] = env_solved(e, c);
            }
        gen += m;
        expanded += npop;
        H.eval(ch.data(), m, D, hv.data());  // all children, before dedup (astar.py:276-278)
        for (int64_t j = 0; j < m; j++) {
            const uint8_t* c = &ch[(size_t)j * D];
            double gp = nodes[popped[j / A]].g + 1.0;  // path_cost + transition cost (astar.py:125-126)
            double hh = std::max((double)hv[j], 0.0);  // clip_zero (nnet_utils.py:193-194)
            double cost = w * gp + hh * (sv[j] ? 0.0 : 1.0);
            auto& bucket = closed_buckets[hash64(c, D)];
            std::pair<int64_t, double>* found = nullptr;
            for (auto& pr : bucket)
                if (memcmp(&pool[(size_t)pr.first * D], c, (size_t)D) == 0) {
                    found = &pr;
                    break;
                }
            bool keep = false;
            if (!found) {
                keep = true;
            } else if (found->second > gp) {
                keep = true;
            }
            if (keep) {
                int64_t nid = (int64_t)nodes.size();
                pool.insert(pool.end(), c, c + D);
                nodes.push_back({gp, cost, popped[j / A], (int8_t)(j % A), (bool)sv[j]});
                if (!found) {
                    bucket.push_back({nid, gp});
                    closed_size++;
                } else {
                    found->second = gp;
                }
                open.push({cost, cnt++, nid});
            }
        }
        if (trace && it < trace_cap) {
            trace[it * 3 + 0] = (int64_t)open.size();
            trace[it * 3 + 1] = closed_size;
            trace[it * 3 + 2] = gen;
        }
        it++;
    }
    res->seconds = now_s() - t0;
    res->nodes_generated = gen;
    res->iterations = it;
    res->nodes_expanded = expanded;
    res->open_size = (int64_t)open.size();
    res->closed_size = closed_size;
    res->solved = !goals.empty();
    res->num_moves = 0;
    res->path_cost = NAN;
    if (!goals.empty()) {
        // get_goal_node_smallest_path_cost (astar.py:327-333): argmin g, first on ties
        int64_t best = goals[0];
        for (int64_t id : goals)
            if (nodes[id].g < nodes[best].g) best = id;
        std::vector<int32_t> mv;
        for (int64_t n = best; nodes[n].parent >= 0; n = nodes[n].parent) mv.push_back(nodes[n].move);
        std::reverse(mv.begin(), mv.end());
        res->num_moves = (int32_t)mv.size();
        res->path_cost = nodes[best].g;
        for (int i = 0; i < (int)mv.size() && i < moves_cap; i++) moves_out[i] = mv[i];
    }
    return 0;
}

/* ==========================================================================================
 * BWAS, cpp semantics — cpp/parallel_weighted_astar.cpp:138-346.  float32 costs
 * (h*!solved + w*depth, :298), std::priority_queue ordered by cost only (:114-119; the same
 * libstdc++ heap the reference uses, driven by the same push/pop sequence), root pushed with
 * cost 0 and a copy inserted in CLOSED (:160-162), pops break at the first solved node
 * (:185-204), deferred termination (:205-208), in-place update of the CLOSED entry when a
 * shallower duplicate appears (:247-265), nodes generated starts at 1 (:166).
 * ========================================================================================== */
struct CNode {
    int64_t state;  // index into pool
    int depth;
    int parentMove;
    float cost;
    float heuristic;
    CNode* parent;
};
struct CCmp {
    bool operator()(const CNode* a, const CNode* b) const { return a->cost > b->cost; }
};

static int astar_cpp(const Env& e, const uint8_t* root, const HeurSrc& H, float depthPenalty, int B,
                     int64_t max_iters, oracle_result* res, int32_t* moves_out, int moves_cap, int64_t* trace,
                     int64_t trace_cap, int stop_on_goal) {
    const int D = e.D, A = e.A;
    std::vector<uint8_t> pool;
    pool.reserve((size_t)1 << 24);
    // stable addresses are not needed: CLOSED is keyed through indices
    struct PKey {
        int64_t idx;
    };
    std::vector<uint8_t>* pp = &pool;
    struct PH {
        std::vector<uint8_t>* pool;
        int D;
        size_t operator()(const PKey& k) const { return (size_t)hash64(&(*pool)[(size_t)k.idx * D], D); }
    };
    struct PE {
        std::vector<uint8_t>* pool;
        int D;
        bool operator()(const PKey& a, const PKey& b) const {
            return memcmp(&(*pool)[(size_t)a.idx * D], &(*pool)[(size_t)b.idx * D], (size_t)D) == 0;
        }
    };
    std::unordered_map<PKey, CNode*, PH, PE> closed(1024, PH{pp, D}, PE{pp, D});
    std::priority_queue<CNode*, std::vector<CNode*>, CCmp> open;
    std::vector<CNode*> all;

    pool.insert(pool.end(), root, root + D);
    CNode* r1 = new CNode{0, 0, -1, 0.f, 0.f, nullptr};
    CNode* r2 = new CNode{0, 0, -1, 0.f, 0.f, nullptr};
    all.push_back(r1);
    all.push_back(r2);
    open.push(r1);            // :160
    closed[PKey{0}] = r2;     // :162
    int64_t gen = 1, it = 0, expanded = 0;  // :166
    bool isSolved = false;
    CNode* solvedNode = nullptr;
    std::vector<CNode*> popped, children, toAdd;
    std::vector<int64_t> toAddIdx;
    std::vector<uint8_t> sv;
    std::vector<float> vals;
    double t0 = now_s();
    while (!isSolved && it < max_iters && !open.empty()) {
        int numPop = (int)std::min<int64_t>((int64_t)open.size(), B);
        popped.clear();
        bool prev = solvedNode != nullptr;
        for (int i = 0; i < numPop; i++) {
            CNode* n = open.top();
            popped.push_back(n);
            open.pop();
            if (env_solved(e, &pool[(size_t)n->state * D])) {
                if (B == 1) {
                    solvedNode = n;
                    isSolved = true;
                } else if (!solvedNode || solvedNode->cost > n->cost) {
                    solvedNode = n;
                }
                break;  // :203
            }
        }
        if (prev && popped[0]->cost >= solvedNode->cost) isSolved = true;  // :205-208
        if (!stop_on_goal) {
            isSolved = false;
        }
        // expand (:217-230)
        int64_t m = (int64_t)popped.size() * A;
        size_t base = pool.size() / D;
        pool.resize(pool.size() + (size_t)m * D);
        children.assign((size_t)m, nullptr);
        sv.resize((size_t)m);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < (int64_t)popped.size(); i++)
            for (int a = 0; a < A; a++) {
                int64_t j = i * A + a;
                uint8_t* c = &pool[(base + j) * D];
                env_move(e, &pool[(size_t)popped[i]->state * D], a, c);
                sv[j] = env_solved(e, c);
                children[j] = new CNode{(int64_t)(base + j), popped[i]->depth + 1, a, 0.f, 0.f, popped[i]};
            }
        expanded += (int64_t)popped.size();
        // closed check, sequential (:244-265)
        toAdd.clear();
        toAddIdx.clear();
        for (int64_t j = 0; j < m; j++) {
            CNode* n = children[j];
            auto f = closed.find(PKey{n->state});
            if (f == closed.end()) {
                closed[PKey{n->state}] = n;
                toAdd.push_back(n);
                toAddIdx.push_back(j);
            } else if (f->second->depth > n->depth) {
                f->second->depth = n->depth;
                f->second->parentMove = n->parentMove;
                f->second->parent = n->parent;
                toAdd.push_back(n);
                toAddIdx.push_back(j);
            } else {
                delete n;
                children[j] = nullptr;
            }
        }
        gen += m;  // :266
        vals.resize((size_t)m);
        H.eval(&pool[base * D], m, D, vals.data());  // all children (:237, :275-279)
        for (size_t i = 0; i < toAdd.size(); i++) {
            float v = std::max(vals[toAddIdx[i]], 0.0f);  // server-side clip (astar.py:491 clip_zero=True)
            // :298  values[i]*(!isSolved) + depthPenalty*((float) depth)   (float32)
            float cost = v * (sv[toAddIdx[i]] ? 0.0f : 1.0f) + depthPenalty * (float)toAdd[i]->depth;
            toAdd[i]->cost = cost;
            toAdd[i]->heuristic = v;
            open.push(toAdd[i]);
            all.push_back(toAdd[i]);
        }
        if (trace && it < trace_cap) {
            trace[it * 3 + 0] = (int64_t)open.size();
            trace[it * 3 + 1] = (int64_t)closed.size();
            trace[it * 3 + 2] = gen;
        }
        it++;
    }
    res->seconds = now_s() - t0;
    res->nodes_generated = gen;
    res->iterations = it;
    res->nodes_expanded = expanded;
    res->open_size = (int64_t)open.size();
    res->closed_size = (int64_t)closed.size();
    res->solved = solvedNode != nullptr && isSolved;
    res->num_moves = 0;
    res->path_cost = NAN;
    if (res->solved) {
        std::vector<int32_t> mv;  // :336-341 prints goal->root; astar.py:529-530 reverses
        for (CNode* n = solvedNode; n->depth > 0; n = n->parent) mv.push_back(n->parentMove);
        std::reverse(mv.begin(), mv.end());
        res->num_moves = (int32_t)mv.size();
        res->path_cost = (double)mv.size();  // astar.py:554 sum of unit transition costs
        for (int i = 0; i < (int)mv.size() && i < moves_cap; i++) moves_out[i] = mv[i];
    }
    for (CNode* n : all) delete n;
    return 0;
}

int oracle_astar(int env, int dim, int semantics, const uint8_t* root, int heur_builtin_id, heur_cb cb, void* user,
                 double weight, int batch, int64_t max_iters, int threads, oracle_result* res, int32_t* moves_out,
                 int moves_cap, int64_t* trace, int64_t trace_cap, int stop_on_goal) {
    Env e = make_env(env, dim);
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    HeurSrc H{cb ? -1 : heur_builtin_id, cb, user};
    if (semantics == SEM_PY)
        return astar_py(e, root, H, weight, batch, max_iters, res, moves_out, moves_cap, trace, trace_cap,
                        stop_on_goal);
    return astar_cpp(e, root, H, (float)weight, batch, max_iters, res, moves_out, moves_cap, trace, trace_cap,
                     stop_on_goal);
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
