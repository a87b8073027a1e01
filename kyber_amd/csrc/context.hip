#include "context.h"
#include "hd.h"

#include <string.h>

#include <algorithm>
#include <map>
#include <thread>
#include <vector>

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

int ctx_workspace(DeviceCtx* ctx, int kind, hipStream_t stream, size_t bytes, void** out, bool* grew) {
    if (grew) *grew = false;
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
        if (grew) *grew = true;
    }
    *out = b.p;
    return KYB_OK;
}

// ---- multi-device host calls
static std::mutex g_md_mu;
static std::vector<int> g_md_devs;          // logical shard -> HIP device (entries may repeat)
static size_t g_md_threshold = 16384;       // smaller host batches stay on the caller's device
static thread_local bool t_in_shard = false;

int md_count() {
    if (t_in_shard) return 1;
    std::lock_guard<std::mutex> lk(g_md_mu);
    return g_md_devs.empty() ? 1 : (int)g_md_devs.size();
}
size_t md_threshold() {
    std::lock_guard<std::mutex> lk(g_md_mu);
    return g_md_threshold;
}
void shard_range(size_t n, int rank, int world, size_t* lo, size_t* hi) {
    const size_t base = n / (size_t)world, rem = n % (size_t)world, r = (size_t)rank;
    *lo = r * base + (r < rem ? r : rem);
    *hi = *lo + base + (r < rem ? 1 : 0);
}
std::vector<int> md_devices() {
    if (t_in_shard) return {};
    std::lock_guard<std::mutex> lk(g_md_mu);
    return g_md_devs;
}
int md_run_impl(size_t n, int (*thunk)(void*, int, size_t, size_t), void* arg, const std::vector<int>* on) {
    // ONE snapshot of the device list per call: a concurrent kyb_set_devices() changes later calls, never the width
    // of one in flight (callers that size buffers by the width pass the snapshot they sized them with)
    const std::vector<int> devs = on ? *on : md_devices();
    const int w = (int)devs.size();
    std::vector<int> rcs(w, KYB_OK);
    std::vector<std::string> errs(w);
    std::vector<std::thread> th;
    th.reserve(w);
    for (int s = 0; s < w; s++) {
        th.emplace_back([&, s]() {
            t_in_shard = true;  // the slice runs the ordinary single-device path
            size_t lo, hi;
            shard_range(n, s, w, &lo, &hi);
            if (hipSetDevice(devs[s]) != hipSuccess) {
                rcs[s] = KYB_E_HIP;
                errs[s] = "hipSetDevice failed for shard " + std::to_string(s);
                return;
            }
            if (hi > lo) rcs[s] = thunk(arg, s, lo, hi);
            if (rcs[s]) errs[s] = g_err;
        });
    }
    for (auto& t : th) t.join();
    for (int s = 0; s < w; s++)
        if (rcs[s]) {
            set_error("shard " + std::to_string(s) + " (device " + std::to_string(devs[s]) + "): " + errs[s]);
            return rcs[s];
        }
    return KYB_OK;
}

}  // namespace kyb

namespace kyb {
int ctx_pin_slots(DeviceCtx::StagePool* pool) {
    if (pool->pinned) return KYB_OK;
    for (int i = 0; i < DeviceCtx::NPIN; i++) {
        if (!pool->pin_in[i]) KYB_HIP_CHECK(hipHostMalloc(&pool->pin_in[i], PIN_IN_BYTES, hipHostMallocDefault));
        if (!pool->pin_out[i]) KYB_HIP_CHECK(hipHostMalloc(&pool->pin_out[i], PIN_OUT_BYTES, hipHostMallocDefault));
    }
    pool->pinned = true;
    return KYB_OK;
}
void par_memcpy(void* dst, const void* src, size_t bytes) {
    constexpr size_t MIN_PART = size_t(1) << 20;
    const int parts = (int)std::min<size_t>(4, bytes / MIN_PART);
    if (parts <= 1) {
        memcpy(dst, src, bytes);
        return;
    }
    const size_t per = ((bytes / parts) + 63) & ~size_t(63);
    std::thread th[3];
    for (int t = 1; t < parts; t++) {
        const size_t lo = per * t, hi = t == parts - 1 ? bytes : per * (t + 1);
        th[t - 1] = std::thread([=] { memcpy((uint8_t*)dst + lo, (const uint8_t*)src + lo, hi - lo); });
    }
    memcpy(dst, src, per);
    for (int t = 1; t < parts; t++) th[t - 1].join();
}
}  // namespace kyb

extern "C" {

int kyb_version(void) { return 2; }

int kyb_set_devices(const int* devices, int ndev) {
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess) have = 0;
    if (ndev < 0 || (ndev && !devices)) {
        kyb::set_error("kyb_set_devices: bad argument");
        return KYB_E_ARG;
    }
    for (int i = 0; i < ndev; i++)
        if (devices[i] < 0 || devices[i] >= have) {
            kyb::set_error("kyb_set_devices: no such device");
            return KYB_E_NODEV;
        }
    std::lock_guard<std::mutex> lk(kyb::g_md_mu);
    kyb::g_md_devs.assign(devices, devices + ndev);
    return KYB_OK;
}
int kyb_init_devices(int ndev) {
    if (ndev < 1) {
        kyb::set_error("kyb_init_devices: ndev must be at least 1");
        return KYB_E_ARG;
    }
    std::vector<int> devs(ndev);
    for (int i = 0; i < ndev; i++) devs[i] = i;
    int rc = kyb_set_devices(devs.data(), ndev);
    if (rc) return rc;
    // eager contexts (tables, staging pools) so that the first batch does not pay for them
    return kyb::md_run_impl((size_t)ndev, [](void*, int, size_t, size_t) -> int {
        kyb::DeviceCtx* ctx;
        return kyb::get_ctx(&ctx);
    }, nullptr);
}
int kyb_get_devices(int* out, int cap) {
    std::lock_guard<std::mutex> lk(kyb::g_md_mu);
    const int n = (int)kyb::g_md_devs.size();
    for (int i = 0; i < n && i < cap; i++) out[i] = kyb::g_md_devs[i];
    return n;
}
int kyb_set_shard_threshold(size_t n) {
    std::lock_guard<std::mutex> lk(kyb::g_md_mu);
    kyb::g_md_threshold = n ? n : 1;
    return KYB_OK;
}
void kyb_shard_range(size_t n, int rank, int world, size_t* lo, size_t* hi) {
    if (world < 1 || rank < 0 || rank >= world) {
        *lo = *hi = 0;
        return;
    }
    kyb::shard_range(n, rank, world, lo, hi);
}
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
        for (auto& pool : c->pools) {
            for (int i = 0; i < kyb::DeviceCtx::NSTAGE; i++)
                if (pool.stage[i]) hipFree(pool.stage[i]);
            for (int i = 0; i < 3; i++)
                if (pool.pipe[i]) hipStreamDestroy(pool.pipe[i]);
            for (int i = 0; i < kyb::DeviceCtx::NPIN; i++) {
                if (pool.pin_in[i]) hipHostFree(pool.pin_in[i]);
                if (pool.pin_out[i]) hipHostFree(pool.pin_out[i]);
            }
            if (pool.stream) hipStreamDestroy(pool.stream);
        }
        delete c;
    }
    kyb::g_ctx.clear();
    if (prev >= 0) hipSetDevice(prev);
    return KYB_OK;
}
}
