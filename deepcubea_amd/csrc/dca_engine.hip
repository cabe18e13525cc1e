// dca_engine.hip — device-resident BWAS engine (placeholder until the engine lands in the next commit)
#include "dca_common.h"
using namespace dca;
extern "C" {
#define NOT_YET() do { set_error("engine not built yet"); return DCA_E_STATE; } while (0)
int dca_engine_create(dca_engine**, int, int, double, int, int64_t, int) { NOT_YET(); }
void dca_engine_destroy(dca_engine*) {}
int dca_engine_reset(dca_engine*, const uint8_t*, void*) { NOT_YET(); }
int dca_engine_root_commit(dca_engine*, const float*, void*) { NOT_YET(); }
int dca_engine_pop_expand(dca_engine*, const uint8_t**, const uint8_t**, int64_t*, void*) { NOT_YET(); }
int dca_engine_commit(dca_engine*, const float*, void*) { NOT_YET(); }
int dca_engine_run_builtin(dca_engine*, int, int, void*) { NOT_YET(); }
int dca_engine_status(dca_engine*, dca_status*, void*) { NOT_YET(); }
int dca_engine_solution(dca_engine*, int32_t*, int, int*, double*, void*) { NOT_YET(); }
int dca_engine_phase_ms(dca_engine*, float*) { NOT_YET(); }
}
