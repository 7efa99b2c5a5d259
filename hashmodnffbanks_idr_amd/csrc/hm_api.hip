// hm_api.hip - host-side bookkeeping of the C ABI (error string, level-table descriptor).
#include "hm_common.h"

#include <new>

namespace {
thread_local std::string g_last_error;
}

void hm_set_error(const std::string &msg) { g_last_error = msg; }
int hm_fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

extern "C" {

int hm_version(void) { return 100; }  // 0.1.0

const char *hm_last_error(void) { return g_last_error.c_str(); }

int hm_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return hm_fail(HM_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    return n;
}

int hm_grid_desc_create(int n_levels, int n_features, const int32_t *res, const uint32_t *rows,
                        const uint64_t *row_off, hm_grid_desc **out) {
    HM_CHECK_ARG(out != nullptr, "hm_grid_desc_create: out is NULL");
    *out = nullptr;
    HM_CHECK_ARG(n_levels >= 1 && n_levels <= HM_MAX_LEVELS, "hm_grid_desc_create: n_levels must be in [1, 32]");
    HM_CHECK_ARG(n_features >= 1 && n_features <= 8, "hm_grid_desc_create: n_features must be in [1, 8]");
    HM_CHECK_ARG(res && rows && row_off, "hm_grid_desc_create: NULL array");
    hm_grid_desc *d = new (std::nothrow) hm_grid_desc();
    HM_CHECK_ARG(d != nullptr, "hm_grid_desc_create: out of host memory");
    d->lv.L = n_levels;
    d->lv.F = n_features;
    d->lv.E = 3 + 2 * n_levels + n_levels * n_features;
    d->lv.pad_ = 0;
    for (int l = 0; l < n_levels; ++l) {
        if (res[l] < 1 || rows[l] < 1 || row_off[l] + rows[l] != row_off[l + 1] || row_off[l + 1] > 0xffffffffull) {
            delete d;
            return hm_fail(HM_ERR_INVALID, "hm_grid_desc_create: inconsistent level table at level " +
                                               std::to_string(l));
        }
        d->lv.res[l] = res[l];
        d->lv.rows[l] = rows[l];
        d->lv.row_off[l] = (uint32_t)row_off[l];
        const bool pow2 = (rows[l] & (rows[l] - 1u)) == 0u;
        d->lv.magic[l] = pow2 ? 0u : (uint32_t)((1ull << 32) / rows[l]);
    }
    d->total_rows = row_off[n_levels];
    *out = d;
    return HM_OK;
}

void hm_grid_desc_destroy(hm_grid_desc *desc) { delete desc; }

int hm_grid_embed_dim(const hm_grid_desc *desc) {
    HM_CHECK_ARG(desc != nullptr, "hm_grid_embed_dim: desc is NULL");
    return desc->lv.E;
}

}  // extern "C"
