// gpt_engine.hip -- placeholder until the decode engine lands (every call reports IXTTS_ERR_STATE).
#include "common.h"
using namespace ixtts;
struct ixtts_gpt { int dummy; };
#define NOTYET(name) do { set_error(name ": decode engine not built yet"); return IXTTS_ERR_STATE; } while (0)
extern "C" {
int ixtts_gpt_create(ixtts_gpt**, const ixtts_gpt_cfg*) { NOTYET("gpt_create"); }
int ixtts_gpt_set_tensor(ixtts_gpt*, const char*, const float*, const int64_t*, int) { NOTYET("gpt_set_tensor"); }
int ixtts_gpt_finalize(ixtts_gpt*) { NOTYET("gpt_finalize"); }
int ixtts_gpt_arena(ixtts_gpt*, void**, size_t*) { NOTYET("gpt_arena"); }
int ixtts_gpt_adopt_arena(ixtts_gpt*) { NOTYET("gpt_adopt_arena"); }
int ixtts_gpt_prefill(ixtts_gpt*, int, const float*, int, int, void*) { NOTYET("gpt_prefill"); }
int ixtts_gpt_decode(ixtts_gpt*, int, int, const ixtts_sampler_cfg*, void*) { NOTYET("gpt_decode"); }
int ixtts_gpt_read(ixtts_gpt*, int, int32_t*, int, int*, int*, void*) { NOTYET("gpt_read"); }
int ixtts_gpt_read_logits(ixtts_gpt*, int, float*, void*) { NOTYET("gpt_read_logits"); }
int ixtts_gpt_force_next(ixtts_gpt*, int, int32_t, void*) { NOTYET("gpt_force_next"); }
int ixtts_gpt_latent(ixtts_gpt*, const float*, int, const int32_t*, int, float*, void*) { NOTYET("gpt_latent"); }
int ixtts_gpt_bench_gemv(ixtts_gpt*, int, int, int, void*) { NOTYET("gpt_bench_gemv"); }
double ixtts_gpt_step_bytes(const ixtts_gpt*, int, int) { return 0.0; }
int ixtts_gpt_destroy(ixtts_gpt*) { return IXTTS_OK; }
}
