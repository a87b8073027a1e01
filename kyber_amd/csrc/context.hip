#include "context.h"
#include "hd.h"

#include <map>

static_assert(kyb::FLAG_UNCOMPRESSED == KYB_F_UNCOMPRESSED && kyb::FLAG_UNCOMPRESSED_OUT == KYB_F_UNCOMPRESSED_OUT &&
                  kyb::FLAG_TRUSTED0 == KYB_F_TRUSTED(0),
              "hd.h flag constants must mirror include/kyber_hip.h");
namespace kyb {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

static std::mutex g_mu;
static std::map<int, DeviceCtx*> g_ctx;

int get_ctx(DeviceCtx** out) {
    int dev = 0;
    KYB_HIP_CHECK(hipGetDevice(&dev));
    DeviceCtx* ctx = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_ctx.find(dev);
        if (it == g_ctx.end()) {
            ctx = new DeviceCtx();
            ctx->device = dev;
            g_ctx[dev] = ctx;
        } else {
            ctx = it->second;
        }
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!ctx->ready) {
        hipDeviceProp_t prop;
        KYB_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        ctx->num_cu = prop.multiProcessorCount;
        int rc = ed25519_build_tables(ctx);
        if (rc) return rc;
        ctx->ready = true;
    }
    *out = ctx;
    return KYB_OK;
}

int ctx_workspace(DeviceCtx* ctx, int kind, hipStream_t stream, size_t bytes, void** out) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceCtx::StreamBuf& b = ctx->sws[std::make_pair(kind, stream)];
    if (bytes > b.cap) {
        if (b.p) {
            KYB_HIP_CHECK(hipDeviceSynchronize());
            KYB_HIP_CHECK(hipFree(b.p));
            b.p = nullptr;
            b.cap = 0;
        }
        const size_t want = bytes + bytes / 4 + 256;
        if (hipMalloc(&b.p, want) != hipSuccess) {
            set_error("workspace hipMalloc failed");
            return KYB_E_ALLOC;
        }
        b.cap = want;
    }
    *out = b.p;
    return KYB_OK;
}

}  // namespace kyb

extern "C" {

int kyb_version(void) { return 1; }
const char* kyb_last_error(void) { return kyb::g_err.c_str(); }

int kyb_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int kyb_stream_release(void* stream) {
    int dev = 0;
    KYB_HIP_CHECK(hipGetDevice(&dev));
    kyb::DeviceCtx* ctx = nullptr;
    {
        std::lock_guard<std::mutex> lk(kyb::g_mu);
        auto it = kyb::g_ctx.find(dev);
        if (it == kyb::g_ctx.end()) return KYB_OK;  // nothing was ever allocated on this device
        ctx = it->second;
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    bool synced = false;
    for (auto it = ctx->sws.begin(); it != ctx->sws.end();) {
        if (it->first.second != (hipStream_t)stream) {
            ++it;
            continue;
        }
        if (it->second.p) {
            if (!synced) {
                KYB_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));  // work enqueued on it may still use the buffer
                synced = true;
            }
            KYB_HIP_CHECK(hipFree(it->second.p));
        }
        it = ctx->sws.erase(it);
    }
    return KYB_OK;
}

int kyb_init(void) {
    kyb::DeviceCtx* ctx;
    return kyb::get_ctx(&ctx);
}

int kyb_shutdown(void) {
    std::lock_guard<std::mutex> lk(kyb::g_mu);
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;  // the caller's current device is restored below
    for (auto& kv : kyb::g_ctx) {
        kyb::DeviceCtx* c = kv.second;
        hipSetDevice(c->device);
        kyb::ed25519_free_tables(c);
        for (auto& w : c->sws)
            if (w.second.p) hipFree(w.second.p);
        for (int i = 0; i < kyb::DeviceCtx::NSTAGE; i++)
            if (c->stage[i]) hipFree(c->stage[i]);
        for (int i = 0; i < 3; i++)
            if (c->pipe[i]) hipStreamDestroy(c->pipe[i]);
        delete c;
    }
    kyb::g_ctx.clear();
    if (prev >= 0) hipSetDevice(prev);
    return KYB_OK;
}
}
