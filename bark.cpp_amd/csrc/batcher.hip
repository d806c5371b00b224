// batcher.hip - a native request collector in front of bark_hip_generate_batch.
//
// The reference's HTTP example serialises its requests on one mutex around bark_generate_audio (examples/server/server.cpp:76-94,
// 128-163): one utterance in flight, every decode step streams the weights for a single sequence.  This component is what a server puts
// in that place on this engine: any number of host threads submit texts; ONE worker thread owns the context, collects what is pending
// (up to max_batch requests, waiting at most max_wait_ms for a batch to fill once the first request is there) and runs them as one
// lock-step batch.  Per-request results are those of a fresh context seeded with the request's seed (engine_generate_batch's contract),
// whatever batch a request happened to travel in.  Plain C++ threads; nothing here touches the device.
// Job streams (bark_hip_batcher_create_ex, n_streams 1 .. 4): further workers on clones of the context (own streams, caches and graphs, the
// same weights) serve the same queue - a lock step is a chain of ~100 small dependent kernels, so two jobs share the chip (two streams of
// 64-slot jobs out of phase: 37.8 prompts/s against 33.2 from one, profiles/r04_staggered_jobs.txt).  Workers that start together stay in
// phase, so the first job of worker i > 0 is capped at max_batch / 2: the streams then run about half a job apart.
#include "engine_internal.h"

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>

using namespace barkhip;

struct bark_hip_batcher {
    struct Req { std::string text; bark_hip_request_params rp{}; std::vector<float> pcm; bool done = false, ok = false; };
    bark_context * ctx = nullptr;                          // worker 0's context (the caller's); request defaults are read from it
    std::vector<bark_context *> ctxs;                      // one per worker; ctxs[1 ..] are clones owned by the batcher ...
    bool owns_workers = true;                              // ... unless the caller brought every context itself (bark_hip_batcher_create_multi: one per GPU)
    int max_batch = 32;
    std::chrono::microseconds max_wait{2000};
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<std::pair<int64_t, std::shared_ptr<Req>>> queue;
    std::unordered_map<int64_t, std::shared_ptr<Req>> tickets;
    int64_t next_ticket = 1;
    bool stop = false;
    int n_batches = 0, n_requests = 0, largest = 0, n_admitted = 0;      // n_admitted: requests that joined a running job
    std::vector<std::thread> workers;

    void run(int wi) {
        bark_context * ctx = ctxs[(size_t) wi];
        bool first = wi > 0 && owns_workers;              // job streams on ONE GPU are kept out of phase (file header); workers on GPUs of their own need not be
        std::unique_lock<std::mutex> lk(mu);
        while (true) {
            cv_work.wait(lk, [&] { return stop || !queue.empty(); });
            if (stop && queue.empty()) return;
            // the first request is there: give the batch max_wait to fill (a full batch leaves at once)
            const auto deadline = std::chrono::steady_clock::now() + max_wait;
            cv_work.wait_until(lk, deadline, [&] { return stop || (int) queue.size() >= max_batch; });
            const int job_cap = first ? std::max(1, max_batch / 2) : max_batch;       // de-phases the job streams (see the file header)
            first = false;
            std::vector<std::shared_ptr<Req>> batch;
            while (!queue.empty() && (int) batch.size() < job_cap) { batch.push_back(queue.front().second); queue.pop_front(); }
            if (batch.empty()) continue;                         // another worker took what was there
            lk.unlock();
            std::vector<const char *> texts; std::vector<bark_hip_request_params> rps;
            for (auto & r : batch) { texts.push_back(r->text.c_str()); rps.push_back(r->rp); }
            // continuous admission: requests that arrive while the job's semantic stage has free slots join it (engine_generate_batch asks here)
            BatchAdmit admit;
            admit.max_job = job_cap;
            admit.next = [&](std::string & text, bark_hip_request_params & rp) {
                std::lock_guard<std::mutex> g(mu);
                if (queue.empty() || (int) batch.size() >= job_cap) return false;
                batch.push_back(queue.front().second); queue.pop_front();
                text = batch.back()->text; rp = batch.back()->rp;
                n_admitted++;
                return true;
            };
            bool failed = false;
            try { engine_generate_batch(ctx, texts.data(), (int) texts.size(), nullptr, rps.data(), &admit); }
            catch (const std::exception & e) { fprintf(stderr, "bark_hip_batcher: batch failed: %s\n", e.what()); failed = true; }
            lk.lock();
            for (size_t i = 0; i < batch.size(); i++) {
                Req & r = *batch[i];
                if (!failed && i < ctx->batch_results.size() && ctx->batch_results[i].ok) { r.pcm = ctx->batch_results[i].audio; r.ok = true; }
                r.done = true;
            }
            n_batches++; n_requests += (int) batch.size(); largest = std::max(largest, (int) batch.size());
            cv_done.notify_all();
        }
    }
};

extern "C" {

BARK_API struct bark_hip_batcher * bark_hip_batcher_create_ex(struct bark_context * bctx, int max_batch, int max_wait_ms, int n_streams) {
    if (!bctx || max_batch < 1 || max_batch > 256 || max_wait_ms < 0 || n_streams < 1 || n_streams > 4) return nullptr;
    std::unique_ptr<bark_hip_batcher> b(new bark_hip_batcher());
    try {
        b->ctx = bctx; b->max_batch = max_batch; b->max_wait = std::chrono::microseconds((int64_t) max_wait_ms * 1000);
        b->ctxs.push_back(bctx);
        for (int i = 1; i < n_streams; i++) {
            b->ctxs.push_back(engine_clone(bctx, (uint32_t) i));
            // the progress callback belongs to the thread that calls into the context (bark.h contract): the further job streams run on
            // threads of their own, so they report nothing instead of calling the user's function concurrently with the same user_data
            b->ctxs.back()->params.progress_callback = nullptr; b->ctxs.back()->params.progress_callback_user_data = nullptr;
        }
        for (bark_context * c : b->ctxs) engine_reserve_batch(c, std::min(max_batch, 64));   // the slot count is fixed by the first use; a larger job queues for the slots
        bark_hip_batcher * raw = b.get();
        for (int i = 0; i < n_streams; i++) b->workers.emplace_back([raw, i] { raw->run(i); });
        return b.release();
    } catch (const std::exception & e) {
        fprintf(stderr, "bark_hip_batcher_create: %s\n", e.what());
        // workers that did start must be joined before the object goes away (a joinable std::thread's destructor terminates the process)
        { std::lock_guard<std::mutex> lk(b->mu); b->stop = true; b->cv_work.notify_all(); }
        for (auto & w : b->workers) if (w.joinable()) w.join();
        for (size_t i = 1; i < b->ctxs.size(); i++) delete b->ctxs[i];
        return nullptr;
    }
}
// Several GPUs behind ONE queue (north_star: many prompts sharded across the GPUs of a node as an embarrassingly-parallel split, no collective inside an
// utterance - here natively, without torch.distributed): worker i is the caller's context i, typically loaded on device i with
// bark_hip_load_model_on_device; every engine entry point selects its context's device first, so a worker thread needs no device state of its own.
// The collector does not own these contexts (bark_free them after bark_hip_batcher_free).  Request defaults are read from ctxs[0].
BARK_API struct bark_hip_batcher * bark_hip_batcher_create_multi(struct bark_context * const * ctxs, int n_ctx, int max_batch, int max_wait_ms) {
    if (!ctxs || n_ctx < 1 || n_ctx > 64 || max_batch < 1 || max_batch > 256 || max_wait_ms < 0) return nullptr;
    for (int i = 0; i < n_ctx; i++) {
        if (!ctxs[i]) return nullptr;
        for (int j = 0; j < i; j++) if (ctxs[j] == ctxs[i]) return nullptr;            // one worker per context
    }
    std::unique_ptr<bark_hip_batcher> b(new bark_hip_batcher());
    try {
        b->ctx = ctxs[0]; b->max_batch = max_batch; b->max_wait = std::chrono::microseconds((int64_t) max_wait_ms * 1000);
        b->owns_workers = false;
        for (int i = 0; i < n_ctx; i++) b->ctxs.push_back(ctxs[i]);
        for (bark_context * c : b->ctxs) engine_reserve_batch(c, std::min(max_batch, 64));
        bark_hip_batcher * raw = b.get();
        for (int i = 0; i < n_ctx; i++) b->workers.emplace_back([raw, i] { raw->run(i); });
        return b.release();
    } catch (const std::exception & e) {
        fprintf(stderr, "bark_hip_batcher_create_multi: %s\n", e.what());
        { std::lock_guard<std::mutex> lk(b->mu); b->stop = true; b->cv_work.notify_all(); }
        for (auto & w : b->workers) if (w.joinable()) w.join();
        return nullptr;
    }
}
BARK_API struct bark_hip_batcher * bark_hip_batcher_create(struct bark_context * bctx, int max_batch, int max_wait_ms) {
    return bark_hip_batcher_create_ex(bctx, max_batch, max_wait_ms, 1);
}

static int64_t batcher_enqueue(struct bark_hip_batcher * b, const char * text, const bark_hip_request_params & rp) {
    auto r = std::make_shared<bark_hip_batcher::Req>();
    r->text = text; r->rp = rp;
    std::lock_guard<std::mutex> lk(b->mu);
    if (b->stop) return -1;
    const int64_t t = b->next_ticket++;
    b->tickets[t] = r;
    b->queue.emplace_back(t, r);
    b->cv_work.notify_all();
    return t;
}
static bark_hip_request_params context_request_params(const bark_context * c, uint32_t seed) {
    bark_hip_request_params rp{};
    rp.temp = c->params.temp; rp.fine_temp = c->params.fine_temp; rp.min_eos_p = c->params.min_eos_p; rp.n_steps_text_encoder = c->params.n_steps_text_encoder; rp.seed = seed;
    return rp;
}
BARK_API int64_t bark_hip_batcher_submit(struct bark_hip_batcher * b, const char * text, uint32_t seed) {
    if (!b || !text) return -1;
    return batcher_enqueue(b, text, context_request_params(b->ctx, seed));
}
BARK_API int64_t bark_hip_batcher_submit_ex(struct bark_hip_batcher * b, const char * text, const struct bark_hip_request_params * params) {
    if (!b || !text) return -1;
    return batcher_enqueue(b, text, params ? *params : context_request_params(b->ctx, 0));
}

BARK_API int bark_hip_batcher_wait(struct bark_hip_batcher * b, int64_t ticket, float * pcm, int capacity) {
    if (!b) return -1;
    std::unique_lock<std::mutex> lk(b->mu);
    auto it = b->tickets.find(ticket);
    if (it == b->tickets.end()) return -1;
    std::shared_ptr<bark_hip_batcher::Req> r = it->second;
    b->cv_done.wait(lk, [&] { return r->done; });
    if (!r->ok) { b->tickets.erase(ticket); return -1; }
    if (!pcm || capacity < (int) r->pcm.size()) return -2 - (int) r->pcm.size();       // too small: -(2 + samples); the ticket stays valid
    memcpy(pcm, r->pcm.data(), r->pcm.size() * sizeof(float));
    const int n = (int) r->pcm.size();
    b->tickets.erase(ticket);
    return n;
}

BARK_API void bark_hip_batcher_stats(struct bark_hip_batcher * b, int * n_batches, int * n_requests, int * largest_batch) {
    if (!b) return;
    std::lock_guard<std::mutex> lk(b->mu);
    if (n_batches) *n_batches = b->n_batches;
    if (n_requests) *n_requests = b->n_requests;
    if (largest_batch) *largest_batch = b->largest;
}

BARK_API int bark_hip_batcher_admitted(struct bark_hip_batcher * b) {
    if (!b) return -1;
    std::lock_guard<std::mutex> lk(b->mu);
    return b->n_admitted;
}

BARK_API void bark_hip_batcher_free(struct bark_hip_batcher * b) {
    if (!b) return;
    { std::lock_guard<std::mutex> lk(b->mu); b->stop = true; b->cv_work.notify_all(); }
    for (auto & w : b->workers) if (w.joinable()) w.join();    // pending requests are still served
    { std::lock_guard<std::mutex> lk(b->mu); b->tickets.clear(); }   // tickets nobody waited for
    if (b->owns_workers) for (size_t i = 1; i < b->ctxs.size(); i++) delete b->ctxs[i];
    delete b;
}

}  // extern "C"
