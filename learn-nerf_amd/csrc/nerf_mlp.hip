// TEMPORARY STUB (replaced by the fused bf16 MFMA kernels)
#include "common.h"
extern "C" int64_t lnrf_nerf_param_count(const lnrf_nerf_shape*) { return 593924; }
extern "C" int64_t lnrf_nerf_packed_bytes(const lnrf_nerf_shape*) { return 0; }
extern "C" int64_t lnrf_nerf_save_bytes(const lnrf_nerf_shape*, int64_t) { return 0; }
extern "C" int64_t lnrf_nerf_bwd_scratch_bytes(const lnrf_nerf_shape*, int64_t) { return 0; }
extern "C" int lnrf_nerf_pack_weights(const lnrf_nerf_shape*, const float*, void*, lnrf_stream_t) { return LNRF_ERR_UNSUPPORTED; }
extern "C" int lnrf_nerf_mlp_fwd(const lnrf_nerf_shape*, const void*, const float*, const float*, const float*, int64_t, const float*, int32_t, int64_t, float*, float*, void*, lnrf_stream_t) { return LNRF_ERR_UNSUPPORTED; }
extern "C" int lnrf_nerf_mlp_bwd(const lnrf_nerf_shape*, const void*, const void*, const float*, const float*, const float*, const float*, int64_t, void*, float*, lnrf_stream_t) { return LNRF_ERR_UNSUPPORTED; }
