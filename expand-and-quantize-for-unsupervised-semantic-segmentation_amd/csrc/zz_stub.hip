// temporary: entry points not written yet
#include "equss_common.h"
extern "C" int equss_layernorm_bf16(const float*, int64_t, const float*, const float*, void*, int64_t, float*, int, int, int, float, void*) { return equss::fail(EQUSS_E_UNSUPPORTED, "not built yet"); }
extern "C" int equss_gemm_bf16(const void*, int64_t, const void*, int64_t, const float*, void*, int64_t, int, int, int, int, void*) { return equss::fail(EQUSS_E_UNSUPPORTED, "not built yet"); }
extern "C" int equss_attention_bf16(const void*, void*, int, int, int, float, void*) { return equss::fail(EQUSS_E_UNSUPPORTED, "not built yet"); }
extern "C" size_t equss_patch_workspace_bytes(int, int, int, int, int) { return 0; }
extern "C" int equss_patch_tokens(const float*, const void*, const float*, const float*, const float*, float*, void*, size_t, int, int, int, int, int, void*) { return equss::fail(EQUSS_E_UNSUPPORTED, "not built yet"); }
extern "C" size_t equss_vit_workspace_bytes(const EqussVit*, int, int, int) { return 0; }
extern "C" int equss_vit_forward(const EqussVit*, const float*, const float*, void*, int64_t, float*, void*, size_t, int, int, int, void*) { return equss::fail(EQUSS_E_UNSUPPORTED, "not built yet"); }
extern "C" int equss_expand_head(void*, const void*, const float*, const void*, const float*, float*, int, int, int, void*) { return equss::fail(EQUSS_E_UNSUPPORTED, "not built yet"); }
