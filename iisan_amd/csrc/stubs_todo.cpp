// TEMPORARY: entry points not implemented yet return an error (never a silent fallback).
#include "common.h"
#define STUB(name, ...) extern "C" int name(__VA_ARGS__) { iisan_set_error(#name ": not implemented yet"); return IISAN_EBADSHAPE; }
extern "C" size_t iisan_side_net_ws_bytes(const iisan_side_cfg*, int64_t) { return 0; }
STUB(iisan_side_net_fwd, const iisan_side_cfg*, const float*, const float*, int64_t, const void* const*, float*, void*, size_t, void*)
STUB(iisan_side_net_bwd, const iisan_side_cfg*, const float*, const float*, int64_t, const void* const*, const float*, void* const*, void*, size_t, void*)
STUB(iisan_linear_fwd, const float*, const float*, const float*, float*, int64_t, int32_t, int32_t, void*)
STUB(iisan_linear_bwd, const float*, const float*, const float*, float*, float*, float*, int64_t, int32_t, int32_t, void*)
extern "C" size_t iisan_sasrec_ws_bytes(const iisan_sasrec_cfg*, int64_t) { return 0; }
STUB(iisan_sasrec_fwd, const iisan_sasrec_cfg*, const float*, const float*, int64_t, const void* const*, float*, void*, size_t, void*)
STUB(iisan_sasrec_bwd, const iisan_sasrec_cfg*, const float*, const float*, int64_t, const void* const*, const float*, float*, void* const*, void*, size_t, void*)
extern "C" size_t iisan_inbatch_ce_ws_bytes(int64_t, int32_t) { return 0; }
STUB(iisan_inbatch_ce_fwd, const int64_t*, const float*, const float*, const float*, const float*, int64_t, int32_t, int32_t, float*, void*, size_t, void*)
STUB(iisan_inbatch_ce_bwd, const int64_t*, const float*, const float*, const float*, const float*, int64_t, int32_t, int32_t, float, float*, float*, void*, size_t, void*)
STUB(iisan_score_rank, const float*, const float*, int64_t, int64_t, int32_t, const int32_t*, int32_t, const int32_t*, int32_t*, void*)
STUB(iisan_adam_step, float*, const float*, float*, float*, int64_t, const int64_t*, const float*, int32_t, int32_t, float, float, float, float, void*)
STUB(iisan_gemm32, const float*, const float*, const float*, float*, int64_t, int32_t, int64_t, int32_t, int32_t, int32_t, int32_t, void*)
