// Native cross-request batching of the request threads' small calls (ABI 14): mq_queue_* — host code only, no kernels of its own.
//
// What it stands in for: the reference serves up to 8 indexing + 8 search request threads at once (/root/reference/src/marqo/api/configs.py:27-28), each
// calling s2_inference.vectorise() with a handful of texts (a search query: ONE — src/marqo/tensor_search/tensor_search.py's query vectorisation; a
// PER_DOCUMENT add_documents: the chunks of one field — src/marqo/core/inference/tensor_fields_container.py:179-223).  Run one by one a tower pass of 1-4
// sequences keeps a few CUs busy; merged, 16 callers' sequences cost about what one caller's do.  marqo_amd/s2_inference/coalesce.py does that merge in Python:
// its leader / follower hand-offs, the tokenised-batch assembly and the torch calls of the merged launch all run under the interpreter lock, which the
// measured merged call pays for (0.92 ms alone, 2.05 ms with 15 other request threads in the interpreter: profiles/r03o_coalesce_profile.txt).  Here the
// merge, the staging and the ~100 launches of the tower pass run on worker threads that never touch the interpreter: a caller blocks inside ONE foreign call
// (ctypes / cgo / JNI release their runtime's lock around it) and wakes up with its rows in its own buffer.
//
// Batching is "natural": a worker that finds requests waiting takes as many as fit one tower call (max_seqs sequences / max_rows token rows) and runs them
// as ONE mq_encode_clip_text / mq_encode_bert on its own stream; whatever arrives meanwhile waits for the next free worker, i.e. groups grow with the load
// and a lone caller is never delayed.  `depth` workers (lanes; each with its own stream, device scratch and pinned staging).  Equal lanes cost heavy load its large
// groups (one worker forms the largest ones and a small-row tower call is host-launch-bound whatever its size — depth 1 > 4 > 3 > 2 measured,
// profiles/r08d_queue_depth_window_sweep.txt), one lane makes two or three callers wait for each other: with helper_seqs > 0 the lanes beyond the first work
// only under light load (lane_run).  window_us > 0: a non-helper lane holds a group that is not full back that long while another call is executing.  A group of ONE sequence replays a hipGraph of its token count (lane_graph_one).  MQ_QUEUE_IMAGE_F32: the same in front of
// mq_encode_image_f32, a request = device addresses of preprocessed images, gathered per group into the lane's batch buffer.
// Rows of a batch are independent in these towers (per-row normalisation, no cross-sequence reduction): a request's embeddings are those of the merged call,
// bit-identical to a lone call of the same kernel family (the small-row families take over at <= 320 rows: DESIGN.md section 3, Numerics).
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace {

struct QRequest {
    const int32_t* ids;
    const int32_t* lens;
    const float* const* imgs;   // MQ_QUEUE_IMAGE_F32: nseq device pointers to [3, S, S] fp32 images (ids / lens unused)
    int64_t nseq, rows;
    float* out;
    int status = MQ_OK;
    bool done = false;
    std::string err;
    std::chrono::steady_clock::time_point t_in;
};

struct QGraph {           // the captured launch sequence of ONE sequence of a given token count on a lane (its buffers and stream are baked in)
    hipGraphExec_t exec = nullptr;
    int seen = 0;
    bool failed = false;
};
constexpr size_t MAX_GRAPHS_PER_LANE = 160;

struct QLane {            // one worker: everything a merged call touches is its own
    hipStream_t stream = nullptr;
    int32_t cap_seqs = 0, cap_rows = 0;          // what this lane's buffers hold: the queue's limits, or a helper lane's small share
    std::unordered_map<int, QGraph> graphs;
    int32_t *d_ids = nullptr, *d_cu = nullptr;
    float* d_in = nullptr;                       // MQ_QUEUE_IMAGE_F32: the gathered batch [max_seqs, 3, S, S]
    float* d_out = nullptr;
    void* d_ws = nullptr;
    size_t ws_bytes = 0;
    int32_t *h_ids = nullptr, *h_cu = nullptr;   // pinned
    float* h_out = nullptr;                      // pinned
    std::thread th;
};

}  // namespace

struct mq_queue {
    mq_queue_cfg cfg;
    const void* tower_cfg;
    const void* tower_w;
    int32_t max_len = 0, vocab = 0, out_dim = 0;
    size_t img_elems = 0;        // MQ_QUEUE_IMAGE_F32: 3 * S * S
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<QRequest*> pending;
    int64_t pending_seqs = 0;
    int busy = 0;
    bool lane0_busy = false;     // the first lane is executing a group ...
    int64_t lane0_seqs = 0;      // ... of this many sequences (what the helper lanes judge the load by)
    bool stop = false;
    std::vector<QLane> lanes;
    mq_queue_stats st{};
};

namespace {

void lane_free(QLane& ln) {
    for (auto& kv : ln.graphs)
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    ln.graphs.clear();
    if (ln.stream) (void)hipStreamDestroy(ln.stream);
    if (ln.d_ids) (void)hipFree(ln.d_ids);
    if (ln.d_cu) (void)hipFree(ln.d_cu);
    if (ln.d_in) (void)hipFree(ln.d_in);
    if (ln.d_out) (void)hipFree(ln.d_out);
    if (ln.d_ws) (void)hipFree(ln.d_ws);
    if (ln.h_ids) (void)hipHostFree(ln.h_ids);
    if (ln.h_cu) (void)hipHostFree(ln.h_cu);
    if (ln.h_out) (void)hipHostFree(ln.h_out);
    ln = QLane{};
}

int lane_alloc(mq_queue* q, QLane& ln, bool helper) {
    // a helper lane only ever runs groups of at most helper_seqs sequences: its scratch is sized for those (a 512-position tower: 2 k rows instead of 16 k)
    const bool image = q->cfg.kind == MQ_QUEUE_IMAGE_F32;
    ln.cap_seqs = helper ? q->cfg.helper_seqs : q->cfg.max_seqs;
    ln.cap_rows = image ? ln.cap_seqs : helper ? (int32_t)std::min<int64_t>(q->cfg.max_rows, (int64_t)q->cfg.helper_seqs * q->max_len) : q->cfg.max_rows;
    const size_t rows = (size_t)ln.cap_rows, seqs = (size_t)ln.cap_seqs, D = (size_t)q->out_dim;
    ln.ws_bytes = image ? mq_vit_workspace_bytes((const mq_vit_cfg*)q->tower_cfg, (int64_t)seqs)
                  : q->cfg.kind == MQ_QUEUE_CLIP_TEXT ? mq_clip_text_workspace_bytes((const mq_clip_text_cfg*)q->tower_cfg, (int64_t)rows, (int64_t)seqs)
                                                      : mq_bert_workspace_bytes((const mq_bert_cfg*)q->tower_cfg, (int64_t)rows, (int64_t)seqs);
    MQ_CHECK_ARG(ln.ws_bytes > 0, "mq_queue_create: the tower reports no workspace for %zu rows / %zu sequences (bad tower cfg?)", rows, seqs);
    ln.ws_bytes += 256;
    MQ_CHECK_HIP(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
    if (image) {
        MQ_CHECK_HIP(hipMalloc((void**)&ln.d_in, seqs * q->img_elems * 4));
    } else {
        MQ_CHECK_HIP(hipMalloc((void**)&ln.d_ids, rows * 4));
        MQ_CHECK_HIP(hipMalloc((void**)&ln.d_cu, (seqs + 1) * 4));
        MQ_CHECK_HIP(hipHostMalloc((void**)&ln.h_ids, rows * 4, hipHostMallocDefault));
        MQ_CHECK_HIP(hipHostMalloc((void**)&ln.h_cu, (seqs + 1) * 4, hipHostMallocDefault));
    }
    MQ_CHECK_HIP(hipMalloc((void**)&ln.d_out, seqs * D * 4));
    MQ_CHECK_HIP(hipMalloc(&ln.d_ws, ln.ws_bytes));
    MQ_CHECK_HIP(hipHostMalloc((void**)&ln.h_out, seqs * D * 4, hipHostMallocDefault));
    return MQ_OK;
}

// H2D of the staged ids / cu_seqlens, the tower pass, D2H of the rows: everything one call ENQUEUES on the lane's stream (eagerly, or into a capture)
int lane_enqueue(mq_queue* q, QLane& ln, int64_t nseq, int64_t rows) {
    const size_t D = (size_t)q->out_dim;
    if (q->cfg.kind == MQ_QUEUE_IMAGE_F32) {   // (the gather into d_in went ahead on this stream: its sources differ from call to call, so it is never part of a graph)
        MQ_TRY(mq_encode_image_f32((const mq_vit_cfg*)q->tower_cfg, (const mq_vit_weights*)q->tower_w, ln.d_in, nseq, ln.d_out, q->cfg.normalize, ln.d_ws,
                                   ln.ws_bytes, ln.stream));
    } else {
        MQ_CHECK_HIP(hipMemcpyAsync(ln.d_ids, ln.h_ids, (size_t)rows * 4, hipMemcpyHostToDevice, ln.stream));
        MQ_CHECK_HIP(hipMemcpyAsync(ln.d_cu, ln.h_cu, (size_t)(nseq + 1) * 4, hipMemcpyHostToDevice, ln.stream));
        if (q->cfg.kind == MQ_QUEUE_CLIP_TEXT)
            MQ_TRY(mq_encode_clip_text((const mq_clip_text_cfg*)q->tower_cfg, (const mq_clip_text_weights*)q->tower_w, ln.d_ids, ln.d_cu, ln.h_cu, nseq, nullptr,
                                       ln.d_out, q->cfg.normalize, ln.d_ws, ln.ws_bytes, ln.stream));
        else
            MQ_TRY(mq_encode_bert((const mq_bert_cfg*)q->tower_cfg, (const mq_bert_weights*)q->tower_w, ln.d_ids, ln.d_cu, ln.h_cu, nseq, ln.d_out,
                                  q->cfg.normalize, ln.d_ws, ln.ws_bytes, ln.stream));
    }
    MQ_CHECK_HIP(hipMemcpyAsync(ln.h_out, ln.d_out, (size_t)nseq * D * 4, hipMemcpyDeviceToHost, ln.stream));
    return MQ_OK;
}

// A group of ONE sequence (the search path's lone query) is ~100 dependent small launches: enqueued eagerly the GPU waits for the host between them
// (~0.45 ms), replayed as a hipGraph it does not (~0.3 ms).  The first call of a token count runs eagerly (one-time host work of the kernels it meets —
// function attributes — happens outside any capture), the second is captured and instantiated, later ones are one hipGraphLaunch.  The lane's staging,
// device buffers and stream are the graph's operands; the tower's cfg / weights are baked in as they are at capture (the Python tower re-creates its queue
// when its policy fields change).  Returns 1 = the rows are in h_out, 0 = not taken (run eagerly), < 0 = error.
int lane_graph_one(mq_queue* q, QLane& ln, int rows) {
    if (!q->cfg.graphs) return 0;
    auto it = ln.graphs.find(rows);
    if (it == ln.graphs.end()) {
        if (ln.graphs.size() >= MAX_GRAPHS_PER_LANE) return 0;
        it = ln.graphs.emplace(rows, QGraph{}).first;
    }
    QGraph& g = it->second;
    if (g.failed) return 0;
    if (!g.exec) {
        if (g.seen++ == 0) return 0;                       // first sighting: eagerly
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(ln.stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { g.failed = true; (void)hipGetLastError(); return 0; }
        const int rc = lane_enqueue(q, ln, 1, rows);
        const hipError_t e = hipStreamEndCapture(ln.stream, &graph);
        if (rc != MQ_OK || e != hipSuccess || !graph || hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0) != hipSuccess) {
            if (graph) (void)hipGraphDestroy(graph);
            g.exec = nullptr;
            g.failed = true;                               // this token count keeps launching eagerly
            (void)hipGetLastError();
            return 0;
        }
        (void)hipGraphDestroy(graph);
        std::lock_guard<std::mutex> lk(q->mu);
        ++q->st.graphs;
    }
    if (hipGraphLaunch(g.exec, ln.stream) != hipSuccess || hipStreamSynchronize(ln.stream) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipGraphExecDestroy(g.exec);
        g.exec = nullptr;
        g.failed = true;
        return 0;
    }
    return 1;
}

// one merged tower call for `group` on the lane's stream; every request's rows are in its own buffer when this returns MQ_OK
int lane_execute(mq_queue* q, QLane& ln, const std::vector<QRequest*>& group) {
    int64_t nseq = 0, rows = 0;
    if (q->cfg.kind == MQ_QUEUE_IMAGE_F32) {
        // the callers' images ([3, S, S] fp32 each, anywhere in HBM: slots of the loaders' preprocess slab) side by side into the lane's batch
        for (const QRequest* r : group)
            for (int64_t i = 0; i < r->nseq; ++i, ++nseq)
                MQ_CHECK_HIP(hipMemcpyAsync(ln.d_in + (size_t)nseq * q->img_elems, r->imgs[i], q->img_elems * 4, hipMemcpyDeviceToDevice, ln.stream));
        rows = nseq;
    } else {
        ln.h_cu[0] = 0;
        for (const QRequest* r : group) {
            for (int64_t s = 0; s < r->nseq; ++s, ++nseq) ln.h_cu[nseq + 1] = ln.h_cu[nseq] + r->lens[s];
            std::memcpy(ln.h_ids + rows, r->ids, (size_t)r->rows * 4);
            rows += r->rows;
        }
    }
    const size_t D = (size_t)q->out_dim;
    const int replayed = nseq == 1 ? lane_graph_one(q, ln, (int)rows) : 0;
    if (replayed < 0) return replayed;
    if (replayed == 1) {
        std::lock_guard<std::mutex> lk(q->mu);
        ++q->st.graph_replays;
    } else {
        MQ_TRY(lane_enqueue(q, ln, nseq, rows));
        MQ_CHECK_HIP(hipStreamSynchronize(ln.stream));
    }
    size_t off = 0;
    for (QRequest* r : group) {
        std::memcpy(r->out, ln.h_out + off * D, (size_t)r->nseq * D * 4);
        off += (size_t)r->nseq;
    }
    return MQ_OK;
}

void lane_run(mq_queue* q, int lane_idx) {
    QLane& ln = q->lanes[(size_t)lane_idx];
    (void)hipSetDevice(q->cfg.device);
    std::vector<QRequest*> group;
    const auto window = std::chrono::microseconds(q->cfg.window_us);
    for (;;) {
        group.clear();
        {
            std::unique_lock<std::mutex> lk(q->mu);
            // helper lanes (helper_seqs > 0: every lane but the first) exist for LIGHT load: two or three request threads never have more than one
            // request waiting, so nothing ever merges and a single lane would run their calls one behind the other (2 threads: 1.0 ms per call where the
            // call itself is 0.5).  A helper takes what is waiting only while that is at most helper_seqs sequences — under heavy load the backlog is larger
            // than that almost always, the helpers sleep and the first lane forms its large groups alone
            // — and only while the group the first lane is running is that small too (a large running group IS heavy load, whatever happens to wait now)
            const bool helper = lane_idx > 0 && q->cfg.helper_seqs > 0;
            auto light = [&] { return q->pending_seqs <= q->cfg.helper_seqs && q->lane0_busy && q->lane0_seqs <= q->cfg.helper_seqs; };
            auto fits = [&](const QRequest* r) { return r->nseq <= ln.cap_seqs && r->rows <= ln.cap_rows; };   // (always, on the first lane)
            q->cv_work.wait(lk, [&] {
                if (q->pending.empty()) return q->stop;
                return fits(q->pending.front()) && (q->stop || !helper || light());   // (a helper drains what fits it when the queue is being destroyed)
            });
            if (q->pending.empty()) return;   // stop, and nothing left to serve
            // company: while another merged call is executing (the GPU is busy anyway) a group that is not full waits until its oldest request is
            // window_us old; arrivals and the other lane's completion wake it
            if (q->cfg.window_us > 0 && !helper) {
                while (!q->stop && q->busy > 0 && !q->pending.empty() && q->pending_seqs < ln.cap_seqs) {
                    const auto deadline = q->pending.front()->t_in + window;
                    if (std::chrono::steady_clock::now() >= deadline) break;
                    q->cv_work.wait_until(lk, deadline);
                }
                if (q->pending.empty()) continue;   // the other lane took them
            }
            int64_t seqs = 0, rows = 0;
            while (!q->pending.empty()) {
                QRequest* r = q->pending.front();
                if (seqs + r->nseq > ln.cap_seqs || rows + r->rows > ln.cap_rows) break;   // (the front request fits: the wait above saw to it)
                group.push_back(r);
                seqs += r->nseq;
                rows += r->rows;
                q->pending.pop_front();
            }
            if (group.empty()) continue;          // (another lane was faster)
            q->pending_seqs -= seqs;
            if (lane_idx == 0) { q->lane0_busy = true; q->lane0_seqs = seqs; }
            ++q->busy;
            ++q->st.calls;
            q->st.requests += (uint64_t)group.size();
            q->st.sequences += (uint64_t)seqs;
            q->st.rows += (uint64_t)rows;
            if ((uint64_t)seqs > q->st.max_call_sequences) q->st.max_call_sequences = (uint64_t)seqs;
            if (group.size() > 1) ++q->st.merged_calls;
        }
        const int rc = lane_execute(q, ln, group);
        std::string err = rc == MQ_OK ? std::string() : std::string(mq_last_error());
        if (rc != MQ_OK) (void)hipStreamSynchronize(ln.stream);   // nothing of a failed call may still be writing when the buffers are reused
        {
            std::lock_guard<std::mutex> lk(q->mu);
            for (QRequest* r : group) {
                r->status = rc;
                r->err = err;
                r->done = true;
            }
            --q->busy;
            if (lane_idx == 0) { q->lane0_busy = false; q->lane0_seqs = 0; }
            if (rc != MQ_OK) ++q->st.failed_calls;
        }
        q->cv_done.notify_all();
        q->cv_work.notify_all();    // a lane holding a group back for company: the GPU is free now
    }
}

}  // namespace

extern "C" int mq_queue_create(const mq_queue_cfg* cfg, const void* tower_cfg, const void* tower_weights, mq_queue** out) {
    MQ_CHECK_ARG(cfg && tower_cfg && tower_weights && out, "mq_queue_create: null pointer");
    *out = nullptr;
    MQ_CHECK_ARG(cfg->kind == MQ_QUEUE_CLIP_TEXT || cfg->kind == MQ_QUEUE_BERT || cfg->kind == MQ_QUEUE_IMAGE_F32,
                 "mq_queue_create: kind %d is none of MQ_QUEUE_CLIP_TEXT / MQ_QUEUE_BERT / MQ_QUEUE_IMAGE_F32", cfg->kind);
    MQ_CHECK_ARG(cfg->max_seqs >= 1 && cfg->max_seqs <= 4096, "mq_queue_create: max_seqs %d outside [1, 4096]", cfg->max_seqs);
    MQ_CHECK_ARG(cfg->depth >= 1 && cfg->depth <= 4, "mq_queue_create: depth %d outside [1, 4]", cfg->depth);
    MQ_CHECK_ARG(cfg->window_us >= 0 && cfg->window_us <= 100000, "mq_queue_create: window_us %d outside [0, 100000]", cfg->window_us);
    MQ_CHECK_ARG(cfg->helper_seqs >= 0 && cfg->helper_seqs <= cfg->max_seqs, "mq_queue_create: helper_seqs %d outside [0, max_seqs]", cfg->helper_seqs);
    int32_t max_len, vocab, out_dim;
    size_t img_elems = 0;
    if (cfg->kind == MQ_QUEUE_IMAGE_F32) {      // a "sequence" is an image, a "row" too: max_rows = max_seqs
        const mq_vit_cfg* t = (const mq_vit_cfg*)tower_cfg;
        max_len = 1;
        vocab = 1;
        out_dim = t->out_dim;
        MQ_CHECK_ARG(t->image_size >= 1 && t->image_size <= 4096, "mq_queue_create: image tower cfg with image_size %d", t->image_size);
        img_elems = (size_t)3 * t->image_size * t->image_size;
    } else if (cfg->kind == MQ_QUEUE_CLIP_TEXT) {
        const mq_clip_text_cfg* t = (const mq_clip_text_cfg*)tower_cfg;
        max_len = t->ctx + (t->cls_pos > 0 ? 1 : 0);
        vocab = t->vocab;
        out_dim = t->out_dim;
    } else {
        const mq_bert_cfg* t = (const mq_bert_cfg*)tower_cfg;
        max_len = t->max_pos;
        vocab = t->vocab;
        out_dim = t->out_dim > 0 ? t->out_dim : t->enc.width;
    }
    MQ_CHECK_ARG(max_len >= 1 && vocab >= 1 && out_dim >= 1, "mq_queue_create: tower cfg without context length / vocabulary / output width");
    MQ_CHECK_ARG(cfg->max_rows >= max_len && (int64_t)cfg->max_rows <= (int64_t)cfg->max_seqs * max_len,
                 "mq_queue_create: max_rows %d outside [longest sequence = %d, max_seqs * that = %ld]", cfg->max_rows, max_len, (long)cfg->max_seqs * max_len);
    int ndev = 0;
    MQ_CHECK_HIP(hipGetDeviceCount(&ndev));
    MQ_CHECK_ARG(cfg->device >= 0 && cfg->device < ndev, "mq_queue_create: device %d of %d", cfg->device, ndev);
    MQ_TRY(mq_check_device(cfg->device));
    int prev = 0;
    MQ_CHECK_HIP(hipGetDevice(&prev));
    MQ_CHECK_HIP(hipSetDevice(cfg->device));
    mq_queue* q = new mq_queue();
    q->cfg = *cfg;
    q->tower_cfg = tower_cfg;
    q->tower_w = tower_weights;
    q->max_len = max_len;
    q->vocab = vocab;
    q->out_dim = out_dim;
    q->img_elems = img_elems;
    q->lanes.resize((size_t)cfg->depth);
    int rc = MQ_OK;
    for (size_t i = 0; i < q->lanes.size(); ++i)
        if ((rc = lane_alloc(q, q->lanes[i], i > 0 && cfg->helper_seqs > 0)) != MQ_OK) break;
    (void)hipSetDevice(prev);
    if (rc != MQ_OK) {
        for (QLane& ln : q->lanes) lane_free(ln);
        delete q;
        return rc;
    }
    for (int i = 0; i < cfg->depth; ++i) q->lanes[(size_t)i].th = std::thread(lane_run, q, i);
    *out = q;
    return MQ_OK;
}

namespace {
// hand a validated request to the workers and block until its rows are in r.out
int submit_and_wait(mq_queue* q, QRequest& r, const char* who) {
    r.t_in = std::chrono::steady_clock::now();
    {
        std::unique_lock<std::mutex> lk(q->mu);
        if (q->stop) {
            mq_set_error("%s: the queue is being destroyed", who);
            return MQ_ERR_INVALID;
        }
        q->pending.push_back(&r);
        q->pending_seqs += r.nseq;
        if (q->lanes.size() > 1) q->cv_work.notify_all();   // (a lane that is holding a group back for company must not swallow the only wake-up)
        else q->cv_work.notify_one();
        q->cv_done.wait(lk, [&] { return r.done; });
    }
    if (r.status != MQ_OK) mq_set_error("%s: the merged tower call failed: %s", who, r.err.c_str());
    return r.status;
}
}  // namespace

extern "C" int mq_queue_encode_images(mq_queue* q, const float* const* d_images, int64_t n, float* h_out) {
    MQ_CHECK_ARG(q, "mq_queue_encode_images: null queue");
    MQ_CHECK_ARG(q->cfg.kind == MQ_QUEUE_IMAGE_F32, "mq_queue_encode_images: this queue serves a text tower (kind %d)", q->cfg.kind);
    if (n == 0) return MQ_OK;
    MQ_CHECK_ARG(d_images && h_out, "mq_queue_encode_images: null buffer");
    MQ_CHECK_ARG(n > 0 && n <= q->cfg.max_seqs, "mq_queue_encode_images: %ld images, the queue takes 1..%d per request", (long)n, q->cfg.max_seqs);
    for (int64_t i = 0; i < n; ++i)
        MQ_CHECK_ARG(d_images[i] && ((uintptr_t)d_images[i] & 15) == 0, "mq_queue_encode_images: image %ld is null or not 16-byte aligned", (long)i);
    QRequest r{};
    r.imgs = d_images;
    r.nseq = n;
    r.rows = n;
    r.out = h_out;
    return submit_and_wait(q, r, "mq_queue_encode_images");
}

extern "C" int mq_queue_encode(mq_queue* q, const int32_t* h_ids, const int32_t* h_lens, int64_t nseq, float* h_out) {
    MQ_CHECK_ARG(q, "mq_queue_encode: null queue");
    MQ_CHECK_ARG(q->cfg.kind != MQ_QUEUE_IMAGE_F32, "mq_queue_encode: this queue serves an image tower (mq_queue_encode_images)");
    if (nseq == 0) return MQ_OK;
    MQ_CHECK_ARG(h_ids && h_lens && h_out, "mq_queue_encode: null buffer");
    MQ_CHECK_ARG(nseq > 0 && nseq <= q->cfg.max_seqs, "mq_queue_encode: %ld sequences, the queue takes 1..%d per request", (long)nseq, q->cfg.max_seqs);
    int64_t rows = 0;
    for (int64_t s = 0; s < nseq; ++s) {
        MQ_CHECK_ARG(h_lens[s] >= 1 && h_lens[s] <= q->max_len, "mq_queue_encode: sequence %ld has %d tokens, the tower takes 1..%d", (long)s, h_lens[s], q->max_len);
        rows += h_lens[s];
    }
    MQ_CHECK_ARG(rows <= q->cfg.max_rows, "mq_queue_encode: %ld token rows, the queue takes up to %d per request", (long)rows, q->cfg.max_rows);
    // an id outside the embedding table would fault on the device and take every request of the merged call with it: refused here, per request
    for (int64_t i = 0; i < rows; ++i) MQ_CHECK_ARG(h_ids[i] >= 0 && h_ids[i] < q->vocab, "mq_queue_encode: token id %d at %ld outside [0, %d)", h_ids[i], (long)i, q->vocab);
    QRequest r{};
    r.ids = h_ids;
    r.lens = h_lens;
    r.nseq = nseq;
    r.rows = rows;
    r.out = h_out;
    return submit_and_wait(q, r, "mq_queue_encode");
}

extern "C" int mq_queue_get_stats(mq_queue* q, mq_queue_stats* out) {
    MQ_CHECK_ARG(q && out, "mq_queue_get_stats: null pointer");
    std::lock_guard<std::mutex> lk(q->mu);
    *out = q->st;
    return MQ_OK;
}

extern "C" int mq_queue_destroy(mq_queue* q) {
    if (!q) return MQ_OK;
    {
        std::lock_guard<std::mutex> lk(q->mu);
        q->stop = true;
    }
    q->cv_work.notify_all();
    for (QLane& ln : q->lanes)
        if (ln.th.joinable()) ln.th.join();   // (workers drain what is still pending: no caller is left blocked)
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(q->cfg.device);
    for (QLane& ln : q->lanes) lane_free(ln);
    (void)hipSetDevice(prev);
    delete q;
    return MQ_OK;
}
