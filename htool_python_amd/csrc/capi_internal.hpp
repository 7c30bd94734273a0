// capi_internal.hpp -- the handle types behind include/htool_mi355x.h, shared by capi.cpp (g++) and dist_device.hip (hipcc)
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/htool_mi355x.h"
#include "hmatrix.hpp"

struct htool_cluster {}; // never instantiated: handles are hm::ClusterHandle
struct htool_generator {
    hm::Generator g;
};
struct DeviceHLU;         // hierarchical LU of an operator, on the device (hlu_device.hip)
struct DeviceDenseFactor; // dense LU / Cholesky of an operator, held on the device (dense_device.hip)
struct htool_hmatrix {
    hm::HMatrix H;
    hm::ClusterHandle *tch = nullptr, *sch = nullptr;
    void *factor = nullptr; // DenseFactor of the host fallback for lu/cholesky (capi.cpp)
    DeviceDenseFactor *dfactor = nullptr; // ... or the device one (larger operators, partition-built blocks)
    struct ::DeviceHLU *hfactor = nullptr;  // the hierarchical factorisation (hlu_device.hip): the default
    ~htool_hmatrix();
};
// dense_device.hip
void device_to_dense_device(const hm::HMatrix &H, void *out_dev, long long ld, void *stream);
bool device_to_dense_host(const hm::HMatrix &H, void *out);
DeviceDenseFactor *device_dense_factor(const hm::HMatrix &H, int kind, char uplo, double shift);
void device_dense_solve(const DeviceDenseFactor *f, char trans, void *B_dev, long long ldb, int mu, void *stream);
void device_dense_solve_host(const hm::HMatrix &H, const DeviceDenseFactor *f, char trans, void *B, int mu);
void device_dense_factor_free(DeviceDenseFactor *f);
int device_dense_factor_kind(const DeviceDenseFactor *f);
// hierarchical LU (hlu.hpp): the plan handle of the diagnostic entries, and the device factorisation (hlu_device.hip)
namespace hm { namespace hlu { struct Plan; } }
struct htool_hlu_plan {
    hm::hlu::Plan *plan = nullptr;
    ~htool_hlu_plan(); // hlu_capi.cpp
};
struct DeviceHLU;
DeviceHLU *device_hlu_factor(const hm::HMatrix &H, int kind, double shift, double eps_lu);
void device_hlu_solve(const DeviceHLU *f, char trans, void *B_dev, long long ldb, int mu, void *stream);
void device_hlu_solve_host(const hm::HMatrix &H, const DeviceHLU *f, char trans, void *B, int mu);
void device_hlu_free(DeviceHLU *f);
int device_hlu_kind(const DeviceHLU *f);
void device_hlu_stats(const DeviceHLU *f, int64_t *out16, double *seconds4);
struct DistDeviceState; // device-side exchange buffers of a distributed operator (dist_device.hip)
struct htool_distributed {
    htool_hmatrix *hmat = nullptr;
    htool_hmatrix *block_diag = nullptr; // (partition rank x partition rank) sub-operator; aliases hmat for one rank
    htool_comm comm;
    const hm::ClusterTree *tc = nullptr, *sc = nullptr;
    std::vector<int64_t> counts, displs;     // rows per rank / first row per rank (cluster numbering, target tree)
    std::vector<int64_t> s_counts, s_displs; // the same for the source tree's partition (empty: it has none of this size)
    DistDeviceState *dev = nullptr;
    // what the deferred build of block_diag needs (it is built when first asked for: utility.hpp:31)
    const htool_generator *gen = nullptr;
    const htool_cluster *t_root = nullptr, *s_root = nullptr;
    htool_build_params params;
    bool block_diag_pending = false;
};

// last error of the calling thread (htool_last_error); set by the API_END macro of every translation unit
std::string &htool_error_slot();

#define API_BEGIN try {
#define API_END                                              \
    }                                                        \
    catch (const std::exception &e) {                        \
        htool_error_slot() = e.what();                       \
        return 1;                                            \
    }                                                        \
    catch (...) {                                            \
        htool_error_slot() = "unknown error";                \
        return 1;                                            \
    }                                                        \
    return 0;

void dist_device_free(DistDeviceState *s); // dist_device.hip
