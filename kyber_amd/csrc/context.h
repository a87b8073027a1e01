// Per-device context shared by every curve's entry points: error reporting,
// lazily built tables and a growable scratch workspace in HBM.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <mutex>
#include <string>

#include "../../include/kyber_hip.h"

namespace kyb {

void set_error(const std::string& msg);

#define KYB_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            ::kyb::set_error(std::string(#expr) + ": " + hipGetErrorString(e_));         \
            return KYB_E_HIP;                                                            \
        }                                                                                \
    } while (0)

struct DeviceCtx {
    int device = -1;
    int num_cu = 0;
    bool ready = false;
    std::mutex mu;
    std::mutex msm_mu;  // serialises host-buffer MSM calls (they share the workspace)
    // Ed25519 fixed-base table: [33][8][3][10] int32 (built on device at init)
    int32_t* ed_base_tab = nullptr;
    // Ed25519 deferred-encoding workspace: parked (X, Y, Z) triples of the last large batch
    void* ed_proj = nullptr;
    size_t ed_proj_bytes = 0;
    // scratch workspace (host entry points stage through it)
    void* ws = nullptr;
    size_t ws_bytes = 0;
};

// Context for the calling thread's current device (created on first use).
int get_ctx(DeviceCtx** out);
// Grow (never shrink) the context's workspace; caller holds no lock.
int ctx_workspace(DeviceCtx* ctx, size_t bytes, void** out);

// Per-curve table builders (defined next to their kernels).
int ed25519_build_tables(DeviceCtx* ctx);
void ed25519_free_tables(DeviceCtx* ctx);

}  // namespace kyb
