// Per-device context shared by every curve's entry points: error reporting,
// lazily built tables and a growable scratch workspace in HBM.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <stdlib.h>

#include <algorithm>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include <utility>

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
    // Held for the whole ENQUEUE sequence of every call that takes a (kind, stream) workspace below -- the ~20 kernels of
    // the MSM pipeline, the Ed25519 batch kernels with their parked triples: two host threads that use the same stream
    // therefore enqueue one whole call after the other, which the stream then runs in that order, and a workspace is
    // never grown (freed) while another thread is still enqueueing kernels on the old pointer.  Since round 6 it is NOT
    // held across a host-buffer call's copies: those calls run on their staging pool's stream (StageScope below), whose
    // workspaces no other call in flight shares.  Recursive: entry points that call other enqueueing helpers re-enter it.
    std::recursive_mutex enq_mu;
    // Ed25519 fixed-base table: [33][136] entries of 32 int32 (30 limbs + 2 pad), built on device at init
    int32_t* ed_base_tab = nullptr;
    // Grow-only device workspaces, one per (kind, stream): calls enqueued on one stream are ordered and reuse their
    // stream's buffer; calls on different streams never share one.  Kinds: WS_MSM (the Pippenger pipeline's arrays),
    // WS_ED (Ed25519 parked (X, Y, Z) triples and per-lane window tables of a large batch).
    struct StreamBuf {
        void* p = nullptr;
        size_t cap = 0;
    };
    std::map<std::pair<int, hipStream_t>, StreamBuf> sws;
    // Staging of the host-buffer entry points: NPOOL pools, each with its own stream, grow-only device buffers (no hipMalloc /
    // hipFree per call), page-locked slots and pipeline streams.  A host-buffer call holds ONE pool for its duration
    // (StageScope) and runs its copies and kernels on the pool's stream, so two host threads on one device overlap one
    // call's staging with the other's kernels (round 6; until then one mutex serialised whole calls per device).  A thread
    // waits when every pool is taken; the lowest free pool is handed out, so a single-threaded caller always meets pool 0 and
    // the per-stream caches behind it (fixed-base tables, a verifier's key lines).  stage_mu guards only the `busy` flags.
    static constexpr int NSTAGE = 8, NPOOL = 2, NPIN = 6;
    struct StagePool {
        bool busy = false;
        hipStream_t stream = nullptr;  // a BLOCKING stream (it synchronises with the null stream, never with the other pool's)
        void* stage[NSTAGE] = {};
        size_t stage_cap[NSTAGE] = {};
        hipStream_t pipe[3] = {};      // chunk-pipelined host paths (H2D | compute | D2H), created on first use
        // page-locked staging of those paths: six chunk slots each way (allocated on first use).  A copy from pageable
        // memory is staged by the runtime at ~10 GB/s and blocks the issuing thread; through these slots the DMA is
        // asynchronous and the host's own memcpy into / out of them overlaps the kernels (ed25519.hip mul_host).
        void* pin_in[NPIN] = {};   // PIN_SLOT_ELEMS * 64 bytes each
        void* pin_out[NPIN] = {};  // PIN_SLOT_ELEMS * 33 bytes each
        bool pinned = false;
    };
    StagePool pools[NPOOL];
    std::mutex stage_mu;
    std::condition_variable stage_cv;
    // fixed_base.cuh: per (workspace kind, stream), the wire bytes + flag bytes of the base whose table the last call
    // through the HOST-buffer entry points enqueued there -- a hint (the chain kernel decides on the device) that lets
    // small same-base batches take the table path.  Guarded by mu; gone with the context at kyb_shutdown.
    std::map<std::pair<int, hipStream_t>, std::string> fb_hint;
};
constexpr size_t PIN_SLOT_ELEMS = size_t(1) << 17;
constexpr size_t PIN_IN_BYTES = PIN_SLOT_ELEMS * 64, PIN_OUT_BYTES = PIN_SLOT_ELEMS * 33;
int ctx_pin_slots(DeviceCtx::StagePool* pool);  // allocates the slots on first use (hipHostMalloc)
// memcpy between pageable buffers and page-locked slots, cut over a few threads: one core moves ~10 GB/s
void par_memcpy(void* dst, const void* src, size_t bytes);

// Flag validation of the pairing-suite entry points: only the documented bits of include/kyber_hip.h for a call with
// `npoint` point arguments (KYB_F_TRUSTED(i), i < npoint), uncompressed input, and -- where the call writes points --
// uncompressed output.  A stray bit (say KYB_F_TRUSTED(2) on a one-point call) is an error, not silently ignored:
// TRUSTED skips checks the GLV / GLS paths rely on.
inline int check_flags(uint32_t flags, int npoint, bool point_out, const char* who) {
    const uint32_t allowed = KYB_F_UNCOMPRESSED | (point_out ? KYB_F_UNCOMPRESSED_OUT : 0u) | (((1u << npoint) - 1u) << 8);
    if (flags & ~allowed) {
        set_error(std::string(who) + ": unknown or inapplicable flag bits");
        return KYB_E_ARG;
    }
    return KYB_OK;
}

// ---- multi-device host calls --------------------------------------------------------
// kyb_set_devices() names the HIP devices the HOST-BUFFER entry points may use.  A call with n units is cut into
// contiguous slices by the shard_range rule (sizes differ by at most one -- the rule kyber_amd/dist.py uses between
// processes), each slice runs on its own host thread bound to its device (own context, staging pool and streams),
// nothing crosses devices except, for an MSM, the encoded partial points that the calling thread adds up at the end.
// The `_dev` entry points are untouched: device pointers belong to the caller's current device.
int md_count();  // shards a host call is cut into; 1 inside a shard and when no device set was given
size_t md_threshold();
void shard_range(size_t n, int rank, int world, size_t* lo, size_t* hi);
// f(shard, lo, hi) -> status code, once per shard, on a thread whose current device is the shard's; returns the first
// non-zero code (its error message becomes the caller's kyb_last_error()).
// `on`: the device list to run on (a snapshot from md_devices()); nullptr = snapshot now.
std::vector<int> md_devices();
int md_run_impl(size_t n, int (*thunk)(void*, int, size_t, size_t), void* arg, const std::vector<int>* on = nullptr);
template <class F>
int md_run(size_t n, F&& f, const std::vector<int>* on = nullptr) {
    return md_run_impl(n, [](void* a, int s, size_t lo, size_t hi) -> int { return (*static_cast<F*>(a))(s, lo, hi); }, &f, on);
}
inline bool md_active(size_t n) { return md_count() > 1 && n >= md_threshold() && n >= (size_t)md_count(); }

// Context for the calling thread's current device (created on first use).
int get_ctx(DeviceCtx** out);
enum { WS_MSM = 0, WS_ED = 1, WS_PAIR = 2, WS_LVM = 3, WS_SCALAR = 4, WS_VKEY = 5, WS_G2TAB = 6, WS_G1TAB = 7, WS_FB = 16 };  // WS_G2TAB: pairing_abi.cuh, the BN G2 ladders' table slabs;  // WS_FB + 2 * suite + group: fixed_base.cuh
// Grow (never shrink) the (kind, stream) workspace; caller holds no lock.
// `grew` (optional): set when the buffer was (re)allocated by this call -- its contents are undefined
int ctx_workspace(DeviceCtx* ctx, int kind, hipStream_t stream, size_t bytes, void** out, bool* grew = nullptr);

// fixed-base table hints (DeviceCtx::fb_hint): set after a successful enqueue, cleared (key == nullptr) by any call that
// may replace the table without the host knowing its base
inline void fb_hint_set(DeviceCtx* ctx, int kind, hipStream_t stream, const std::string* key) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (key) ctx->fb_hint[std::make_pair(kind, stream)] = *key;
    else ctx->fb_hint.erase(std::make_pair(kind, stream));
}
inline bool fb_hint_is(DeviceCtx* ctx, int kind, hipStream_t stream, const std::string& key) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto it = ctx->fb_hint.find(std::make_pair(kind, stream));
    return it != ctx->fb_hint.end() && it->second == key;
}

// A host-buffer call: holds one of the device's staging pools for its duration and hands out its buffers in order.
// KYB_STAGE_POOLS=1 keeps every call on pool 0 (the serialised behaviour of rounds 1-5, for A/B runs).
struct StageScope {
    DeviceCtx* ctx;
    DeviceCtx::StagePool* pool = nullptr;
    int next = 0;
    explicit StageScope(DeviceCtx* c) : ctx(c) {
        static const int npool = [] {
            const char* e = getenv("KYB_STAGE_POOLS");
            const int v = e ? atoi(e) : 0;
            return v >= 1 && v <= DeviceCtx::NPOOL ? v : DeviceCtx::NPOOL;
        }();
        {
            std::unique_lock<std::mutex> lk(c->stage_mu);
            for (;;) {
                for (int i = 0; i < npool && !pool; i++)
                    if (!c->pools[i].busy) pool = &c->pools[i];
                if (pool) break;
                c->stage_cv.wait(lk);
            }
            pool->busy = true;
        }
        // (a stream that cannot be created leaves the pool on the null stream: correct, without the overlap)
        if (!pool->stream && hipStreamCreateWithFlags(&pool->stream, hipStreamDefault) != hipSuccess) pool->stream = nullptr;
        current() = this;
    }
    ~StageScope() {
        current() = nullptr;
        {
            std::lock_guard<std::mutex> lk(ctx->stage_mu);
            pool->busy = false;
        }
        ctx->stage_cv.notify_one();
    }
    StageScope(const StageScope&) = delete;
    StageScope& operator=(const StageScope&) = delete;
    hipStream_t stream() const { return pool->stream; }
    static StageScope*& current() {
        static thread_local StageScope* cur = nullptr;
        return cur;
    }
};
// One device staging buffer of the current StageScope's pool (freed only by kyb_shutdown).  Copies run on the pool's stream;
// download() leaves the stream idle (the call's kernels were enqueued on it: what it copies out is final).
struct StageBuf {
    void* p = nullptr;
    int alloc(size_t bytes) {
        StageScope* sc = StageScope::current();
        if (!sc || sc->next >= DeviceCtx::NSTAGE) {
            set_error("staging: no scope / too many buffers");
            return KYB_E_ARG;
        }
        const int slot = sc->next++;
        DeviceCtx::StagePool* c = sc->pool;
        if (c->stage_cap[slot] < bytes || !c->stage[slot]) {
            if (c->stage[slot]) {
                hipFree(c->stage[slot]);
                c->stage[slot] = nullptr;
                c->stage_cap[slot] = 0;
            }
            const size_t cap = bytes + bytes / 4 + 256;
            if (hipMalloc(&c->stage[slot], cap) != hipSuccess) {
                set_error("hipMalloc failed");
                return KYB_E_ALLOC;
            }
            c->stage_cap[slot] = cap;
        }
        p = c->stage[slot];
        return KYB_OK;
    }
    // Large transfers bounce through two page-locked slots: the runtime stages a copy from / to pageable memory itself
    // at ~10 GB/s and blocks; here a piece is memcpy'd by four host threads while the previous piece moves by DMA
    // (a 2^20-point MSM uploads 84-168 MB -- at 10 GB/s that is longer than the MSM takes).
    static constexpr size_t BOUNCE_MIN = size_t(4) << 20;
    int upload(const void* src, size_t bytes) {
        int rc = alloc(bytes);
        if (rc) return rc;
        if (!bytes) return KYB_OK;
        if (bytes < BOUNCE_MIN) {
            const hipStream_t st = StageScope::current()->stream();
            KYB_HIP_CHECK(hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, st));
            KYB_HIP_CHECK(hipStreamSynchronize(st));  // the source is the caller's pageable memory: done with it on return
            return KYB_OK;
        }
        return bounce(const_cast<void*>(src), bytes, true);
    }
    int download(void* dst, size_t bytes) {
        if (!bytes) return KYB_OK;
        if (bytes < BOUNCE_MIN) {
            const hipStream_t st = StageScope::current()->stream();
            KYB_HIP_CHECK(hipMemcpyAsync(dst, p, bytes, hipMemcpyDeviceToHost, st));
            KYB_HIP_CHECK(hipStreamSynchronize(st));
            return KYB_OK;
        }
        return bounce(dst, bytes, false);
    }
    int bounce(void* host, size_t bytes, bool up) {
        StageScope* sc = StageScope::current();
        DeviceCtx::StagePool* c = sc->pool;
        const hipStream_t st = sc->stream();
        int rc = ctx_pin_slots(c);
        if (rc) return rc;
        hipEvent_t ev[2];
        for (int k = 0; k < 2; k++) KYB_HIP_CHECK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
        const size_t piece = PIN_IN_BYTES;
        const size_t npiece = (bytes + piece - 1) / piece;
        rc = KYB_OK;
        for (size_t i = 0; i <= npiece && rc == KYB_OK; i++) {
            // up:   host -> slot(i) by memcpy, then slot(i) -> device by DMA (async); a slot is reused two pieces later
            // down: device -> slot(i) by DMA (async), slot(i - 1) -> host by memcpy meanwhile
            if (up) {
                if (i == npiece) break;
                const size_t off = i * piece, len = std::min(piece, bytes - off);
                if (i >= 2 && hipEventSynchronize(ev[i & 1]) != hipSuccess) rc = KYB_E_HIP;
                par_memcpy(c->pin_in[i & 1], (const uint8_t*)host + off, len);
                if (hipMemcpyAsync((uint8_t*)p + off, c->pin_in[i & 1], len, hipMemcpyHostToDevice, st) != hipSuccess ||
                    hipEventRecord(ev[i & 1], st) != hipSuccess)
                    rc = KYB_E_HIP;
            } else {
                if (i < npiece) {
                    const size_t off = i * piece, len = std::min(piece, bytes - off);
                    if (hipMemcpyAsync(c->pin_in[i & 1], (const uint8_t*)p + off, len, hipMemcpyDeviceToHost, st) != hipSuccess ||
                        hipEventRecord(ev[i & 1], st) != hipSuccess)
                        rc = KYB_E_HIP;
                }
                if (i >= 1 && rc == KYB_OK) {
                    const size_t off = (i - 1) * piece, len = std::min(piece, bytes - off);
                    if (hipEventSynchronize(ev[(i - 1) & 1]) != hipSuccess) rc = KYB_E_HIP;
                    else par_memcpy((uint8_t*)host + off, c->pin_in[(i - 1) & 1], len);
                }
            }
        }
        if (hipStreamSynchronize(st) != hipSuccess) rc = KYB_E_HIP;
        for (int k = 0; k < 2; k++) hipEventDestroy(ev[k]);
        if (rc == KYB_E_HIP) set_error("staging: page-locked bounce copy failed");
        return rc;
    }
};

// Per-curve table builders (defined next to their kernels).
int ed25519_build_tables(DeviceCtx* ctx);
void ed25519_free_tables(DeviceCtx* ctx);

}  // namespace kyb
