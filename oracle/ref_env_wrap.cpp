/*
 * ref_env_wrap.cpp — thin extern "C" driver around the REFERENCE's own C++ environments.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md).  This file contains no reference code: it
 * includes the reference header where it lies (-I/root/reference/cpp) and is linked against
 * /root/reference/cpp/environments.cpp compiled in place by oracle/Makefile (`make ref`).
 * Output goes to oracle/_ref/libref_env.so only (git-ignored, travels to the GPU box).
 *
 * What it exercises (paths relative to forestagostinelli/DeepCubeA):
 *   cpp/environments.cpp:222-243  Cube3::getNextState / getNextStates
 *   cpp/environments.cpp:92-113   PuzzleN::getNextState / getNextStates
 *   cpp/environments.cpp:119-126,249-256  isSolved
 *   cpp/environments.cpp:133-208  LightsOut (move matrix, getNextState, getNextStates, isSolved)
 *   cpp/environments.cpp:263-370  Cube4 (rotateIdxs_old / _new, getNextState, getNextStates, isSolved)
 * driven the way the reference's hot loop drives them
 * (cpp/parallel_weighted_astar.cpp:217-230: `#pragma omp parallel for` over popped nodes,
 *  one heap-allocated Environment per child).
 *
 * cpp/parallel_weighted_astar.cpp itself is NOT buildable in this image: line 30 includes
 * <boost/functional/hash.hpp>, boost is absent, and writing a stand-in header is not allowed.
 */
#include <cstdint>
#include <cstring>
#include <vector>

#include "environments.h"

#ifdef _OPENMP
#include <omp.h>
#endif

static Environment* make_env(int env, int dim, const uint8_t* s, int D) {
    std::vector<uint8_t> v(s, s + D);
    if (env == 0) return new Cube3(v);
    if (env == 2) return new LightsOut(v, (uint8_t)dim);  // cpp/environments.cpp:156-208 (dim 7: moveMat7)
    if (env == 3) return new Cube4(v);                    // cpp/environments.cpp:322-370
    return new PuzzleN(v, (uint8_t)dim);
}
static inline int env_D(int env, int dim) { return env == 0 ? 54 : env == 3 ? 96 : dim * dim; }
static inline int env_A(int env, int dim) { return env == 0 ? 12 : env == 3 ? 24 : env == 2 ? dim * dim : 4; }

extern "C" {

// children [n, A, D]; solved [n*A] (optional)
void ref_expand(int env, int dim, const uint8_t* in, int64_t n, uint8_t* children, uint8_t* solved, int threads) {
    const int D = env_D(env, dim);
    const int A = env_A(env, dim);
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for
    for (int64_t i = 0; i < n; i++) {
        Environment* e = make_env(env, dim, in + i * D, D);
        std::vector<Environment*> ch = e->getNextStates();
        for (int a = 0; a < A; a++) {
            std::vector<uint8_t> st = ch[a]->getState();
            memcpy(children + (i * A + a) * D, st.data(), (size_t)D);
            if (solved) solved[i * A + a] = ch[a]->isSolved();
            delete ch[a];
        }
        delete e;
    }
}

void ref_next_state(int env, int dim, const uint8_t* in, int64_t n, int action, uint8_t* out) {
    const int D = env_D(env, dim);
    for (int64_t i = 0; i < n; i++) {
        Environment* e = make_env(env, dim, in + i * D, D);
        Environment* c = e->getNextState(action);
        std::vector<uint8_t> st = c->getState();
        memcpy(out + i * D, st.data(), (size_t)D);
        delete c;
        delete e;
    }
}

void ref_is_solved(int env, int dim, const uint8_t* in, int64_t n, uint8_t* out) {
    const int D = env_D(env, dim);
    for (int64_t i = 0; i < n; i++) {
        Environment* e = make_env(env, dim, in + i * D, D);
        out[i] = e->isSolved();
        delete e;
    }
}

int ref_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
}
