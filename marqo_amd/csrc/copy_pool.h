// Host-side copy threads shared by the pack routines (csrc/py_stage.cpp: Pillow images; csrc/runtime.hip: mq_host_gather): header-only,
// plain C++17 + pthread_atfork, one pool per shared library that includes it.
#pragma once
#include <pthread.h>

#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace {

// ---- copy threads ----------------------------------------------------------------------------------------------------------------
// A pack is 13-50 MB of memcpy cut into a few ranges.  Starting std::threads per call cost 20-50 us EACH, in sequence, four times per
// 256-image batch (one call per 64-image slice): the workers are kept instead — parked on a condition variable, woken per call.  A fork()ed child
// starts without them (pthread_atfork) and creates its own on first use.
class CopyPool {
  public:
    // run job(k) for k in [0, n) on up to n - 1 workers + the calling thread; returns when all are done.  Callers hold no GIL.
    void run(int n, const std::function<void(int)>& job) {
        if (n <= 1) { if (n == 1) job(0); return; }
        // The workers serve ONE pack at a time.  A second request thread that arrives meanwhile does not queue behind it (with 8 + 8 request
        // threads that would serialise every pack in the process): it copies its ranges itself — concurrent callers are the parallelism then.
        std::unique_lock<std::mutex> call(call_mu_, std::try_to_lock);
        if (!call.owns_lock()) {
            for (int k = 0; k < n; ++k) job(k);
            return;
        }
        {
            std::lock_guard<std::mutex> g(mu_);
            while ((int)workers_.size() < n - 1 && (int)workers_.size() < kMaxWorkers) {
                try { workers_.emplace_back([this] { loop(); }); } catch (...) { break; }
            }
            job_ = &job;
            next_ = 0;
            total_ = n;
            pending_ = n;
            ++epoch_;
        }
        // From here on workers may be executing job(k), which lives in the CALLER's frame: whatever makes this function leave — normally, or by an
        // exception of a mutex / of the job itself — it first withdraws the ranges nobody has claimed and waits for the claimed ones (a
        // caller's catch(...) used to run while workers were still inside its job).
        struct Drain {
            CopyPool* p;
            ~Drain() {
                try {
                    std::unique_lock<std::mutex> g(p->mu_);
                    p->pending_ -= p->total_ - p->next_;
                    p->next_ = p->total_;
                    p->done_.wait(g, [this] { return p->pending_ <= 0; });
                    p->job_ = nullptr;
                } catch (...) {
                }
            }
        } drain{this};
        wake_.notify_all();
        for (;;) {                                                // the caller takes ranges too (and all of them when no worker could be created)
            int k;
            {
                std::lock_guard<std::mutex> g(mu_);
                if (next_ >= total_) break;
                k = next_++;
            }
            try {
                job(k);
            } catch (...) {
                std::lock_guard<std::mutex> g(mu_);
                --pending_;
                throw;
            }
            std::lock_guard<std::mutex> g(mu_);
            --pending_;
        }
    }
    void forget_threads() {                                        // in a fork()ed child: the parent's workers do not exist here
        new (&workers_) std::vector<std::thread>();
        new (&mu_) std::mutex();
        new (&call_mu_) std::mutex();
        new (&wake_) std::condition_variable();
        new (&done_) std::condition_variable();
        job_ = nullptr;
        next_ = total_ = pending_ = 0;
    }

  private:
    static constexpr int kMaxWorkers = 15;
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)>* job;
            int k;
            {
                std::unique_lock<std::mutex> g(mu_);
                wake_.wait(g, [&] { return epoch_ != seen && job_ != nullptr && next_ < total_; });
                if (next_ >= total_) { seen = epoch_; continue; }
                job = job_;
                k = next_++;
                if (next_ >= total_) seen = epoch_;
            }
            (*job)(k);
            std::lock_guard<std::mutex> g(mu_);
            if (--pending_ == 0) done_.notify_all();
        }
    }
    std::mutex mu_, call_mu_;
    std::condition_variable wake_, done_;
    std::vector<std::thread> workers_;
    const std::function<void(int)>* job_ = nullptr;
    int next_ = 0, total_ = 0, pending_ = 0;
    uint64_t epoch_ = 0;
};
CopyPool* g_pool = nullptr;   // (one per translation unit that includes this header: each shared library has its own workers)
void pool_after_fork() { if (g_pool) g_pool->forget_threads(); }
inline CopyPool& mq_copy_pool() {
    static CopyPool* p = [] {
        g_pool = new CopyPool();                                    // (never destroyed: its threads run until the process exits)
        pthread_atfork(nullptr, nullptr, pool_after_fork);
        return g_pool;
    }();
    return *p;
}

}  // namespace
