// DAC decode (autoencoder.py:119-170) — placeholder entry points until the conv kernels land.
#include "../../include/zonos_hip.h"
#include <string>
static thread_local std::string g_dac_err = "DAC decode kernels not built yet";
extern "C" int zn_dac_create(const zn_dac_config*, const zn_dac_tensor*, int32_t, zn_dac* out) { if (out) *out = nullptr; return ZN_ERR_UNSUPPORTED; }
extern "C" int zn_dac_destroy(zn_dac) { return ZN_OK; }
extern "C" const char* zn_dac_last_error(zn_dac) { return g_dac_err.c_str(); }
extern "C" int zn_dac_decode(zn_dac, const int32_t*, int32_t, int32_t, float*, zn_stream) { return ZN_ERR_UNSUPPORTED; }
