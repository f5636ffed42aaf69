#include <stdlib.h>
// Library-wide host plumbing: thread-local error text, ABI/arch info, per-family launch timing.
#include <stdarg.h>
#include <string.h>
#include <mutex>
#include <vector>
#include "common.h"
#include "copy_pool.h"

static thread_local char g_err[512] = "";

void mq_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mq_last_error(void) { return g_err; }
extern "C" int mq_abi_version(void) { return MQ_ABI_VERSION; }
mq_knob mq_xcd_band{getenv("MQ_XCD_BAND") ? atoi(getenv("MQ_XCD_BAND")) : 1};   // row-wise kernels follow the GEMMs' XCD banding (common.h)
extern "C" const char* mq_build_arch(void) { return "gfx950"; }

// ---- the device this library was written for -------------------------------------------------------------------------------------------------
// The persistent GEMM grids, the XCD-aware tile order and the in-kernel tail count on 256 CUs in 8 XCDs (an MI355X in SPX mode); a partitioned
// (DPX / CPX) or different device would silently mis-tile or strand the tail's workgroups — refuse it loudly instead.  mq_check_device(-1) = the
// current device.  The tiled GEMM launchers call mq_device_ok() (cached per device ordinal).
extern "C" int mq_check_device(int device) {
    int dev = device;
    if (dev < 0) MQ_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    MQ_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    const bool arch_ok = strncmp(prop.gcnArchName, "gfx950", 6) == 0;
    if (!arch_ok || prop.multiProcessorCount != 256) {
        mq_set_error("libmarqo_hip is built for one whole MI355X (gfx950, 256 CUs in 8 XCDs); device %d is %s with %d CUs (a DPX / CPX partition?)", dev,
                     prop.gcnArchName, prop.multiProcessorCount);
        return MQ_ERR_UNSUPPORTED;
    }
    return MQ_OK;
}
int mq_device_ok() {
    static std::atomic<uint64_t> checked{0};
    int dev = 0;
    MQ_CHECK_HIP(hipGetDevice(&dev));
    const uint64_t bit = 1ull << (dev & 63);
    if (checked.load(std::memory_order_acquire) & bit) return MQ_OK;
    MQ_TRY(mq_check_device(dev));
    checked.fetch_or(bit, std::memory_order_release);
    return MQ_OK;
}

// ---- profiling -----------------------------------------------------------------------------
namespace {
struct ProfRec { int family; hipEvent_t start, stop; double flops; };
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
bool g_prof_on = false;
}  // namespace

MqProfScope::MqProfScope(int family, hipStream_t s, double flops)
    : family_(family), stream_(s), start_(nullptr), on_(g_prof_on) {
    if (!on_) return;
    hipEvent_t stop;
    if (hipEventCreate(&start_) != hipSuccess || hipEventCreate(&stop) != hipSuccess) { on_ = false; return; }
    (void)hipEventRecord(start_, stream_);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(ProfRec{family_, start_, stop, flops});
}

MqProfScope::~MqProfScope() {
    if (!on_) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto it = g_prof.rbegin(); it != g_prof.rend(); ++it)
        if (it->start == start_) { (void)hipEventRecord(it->stop, stream_); break; }
}

extern "C" int mq_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return MQ_OK;
}

extern "C" int mq_profile_collect(double* ms, int64_t* launches, double* gemm_flops) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < MQ_PROF_FAMILIES; ++i) { if (ms) ms[i] = 0.0; if (launches) launches[i] = 0; }
    if (gemm_flops) *gemm_flops = 0.0;
    int rc = MQ_OK;
    for (auto& r : g_prof) {
        if (hipEventSynchronize(r.stop) != hipSuccess) { rc = MQ_ERR_HIP; mq_set_error("mq_profile_collect: event sync failed"); }
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.start, r.stop) == hipSuccess && r.family >= 0 && r.family < MQ_PROF_FAMILIES) {
            if (ms) ms[r.family] += t;
            if (launches) launches[r.family] += 1;
            if (r.family == 0 && gemm_flops) *gemm_flops += r.flops;
        }
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    g_prof.clear();
    return rc;
}


// ---- host-side staging helper -------------------------------------------------------------------------------------------------
// Gather n host buffers into one (pinned) staging buffer with a few copy threads, in ONE foreign call: the Python loaders release the GIL
// once for the whole pack instead of once per image (with several request threads, every per-image re-acquisition of the GIL waited for
// the interpreter's switch interval: 256-image calls went from 3 ms to 30-55 ms of packing under 4 concurrent callers).
#include <cstring>
#include <thread>
#include <vector>
extern "C" int mq_host_gather(const void* const* h_src, const int64_t* h_bytes, const int64_t* h_dst_off, int64_t n, void* h_dst, int32_t threads) {
    if (n <= 0) return MQ_OK;
    MQ_CHECK_ARG(h_src && h_bytes && h_dst_off && h_dst, "mq_host_gather: null pointer");
    int64_t total = 0;
    for (int64_t i = 0; i < n; ++i) {
        MQ_CHECK_ARG(h_bytes[i] >= 0 && h_dst_off[i] >= 0 && (h_bytes[i] == 0 || h_src[i]), "mq_host_gather: bad item %ld", (long)i);
        total += h_bytes[i];
    }
    int t = threads < 1 ? 1 : (threads > 16 ? 16 : threads);
    if (total < (4 << 20)) t = 1;   // small packs: waking the copy threads costs more than the copy
    auto work = [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i)
            if (h_bytes[i]) memcpy((char*)h_dst + h_dst_off[i], h_src[i], (size_t)h_bytes[i]);
    };
    if (t == 1) { work(0, n); return MQ_OK; }
    // contiguous item ranges of about total / t bytes each, copied by the kept worker threads + the calling thread (copy_pool.h; a worker that
    // cannot be created — pid / thread limits of a container — just means the calling thread copies more)
    std::vector<std::pair<int64_t, int64_t>> ranges;   // (outside the try: the workers run `job` over them, see CopyPool::run)
    const std::function<void(int)> job = [&](int r) { work(ranges[r].first, ranges[r].second); };
    try {
        int64_t lo = 0, acc = 0;
        const int64_t share = (total + t - 1) / t;
        for (int64_t i = 0; i < n; ++i) {
            acc += h_bytes[i];
            if (acc >= share || i == n - 1) {
                ranges.emplace_back(lo, i + 1);
                lo = i + 1;
                acc = 0;
            }
        }
        mq_copy_pool().run((int)ranges.size(), job);
    } catch (...) {   // (allocation failure while setting up: nothing was copied by halves that matter — copy everything here)
        work(0, n);
    }
    return MQ_OK;
}

// same, with the destination's capacity: every item must lie inside [0, dst_bytes)
extern "C" int mq_host_gather_checked(const void* const* h_src, const int64_t* h_bytes, const int64_t* h_dst_off, int64_t n, void* h_dst,
                                      int64_t dst_bytes, int32_t threads) {
    if (n <= 0) return MQ_OK;
    MQ_CHECK_ARG(h_bytes && h_dst_off && dst_bytes >= 0, "mq_host_gather_checked: null pointer");
    for (int64_t i = 0; i < n; ++i)
        MQ_CHECK_ARG(h_bytes[i] >= 0 && h_dst_off[i] >= 0 && h_dst_off[i] <= dst_bytes && h_bytes[i] <= dst_bytes - h_dst_off[i],
                     "mq_host_gather_checked: item %ld (%ld bytes at %ld) leaves the %ld-byte destination", (long)i, (long)h_bytes[i],
                     (long)h_dst_off[i], (long)dst_bytes);
    return mq_host_gather(h_src, h_bytes, h_dst_off, n, h_dst, threads);
}
