// A launch worker: one host thread of the library that issues the LOADER's native step (tgmx_pipeline_step and the events around it)
// on the loader's own stream, so that the calling thread does not pay for those launches (~5 us of host time each on this stack: a TGN
// batch is 18 launches and host-bound once the two chains overlap on the device -- DESIGN.md 3.3c).  Jobs are executed strictly in the
// order they were submitted; the argument blocks are copied at submission, so the caller may reuse them at once.  Nothing here touches
// the caller's stream: ordering between the two streams stays with the events the caller passes in.
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>

#include "common.h"

namespace {
struct Job {
  uint64_t ticket;
  tgmx_pipeline_t pipe;
  tgmx_pipeline_out_t out;
  tgmx_pipeline_post_t post;
  bool has_post;
  long long lo, n;
  uint64_t call;
  hipStream_t stream;
  hipEvent_t wait_ev, record_ev;
};

struct Worker {
  int device = 0;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  std::deque<Job> jobs;
  uint64_t next_ticket = 1, done_ticket = 0;  // tickets complete in order: done_ticket = the last finished one
  std::unordered_map<uint64_t, std::pair<int, std::string>> failed;  // ticket -> (rc, error text) of the jobs that failed
  bool stop = false;

  void run() {
    (void)hipSetDevice(device);
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_job.wait(lk, [&] { return stop || !jobs.empty(); });
        if (jobs.empty()) return;  // (stop, and nothing left to do)
        j = jobs.front();
        jobs.pop_front();
      }
      int rc = TGMX_OK;
      std::string err;
      if (j.wait_ev && hipStreamWaitEvent(j.stream, j.wait_ev, 0) != hipSuccess) {
        rc = TGMX_E_LAUNCH;
        err = "worker: hipStreamWaitEvent failed";
      }
      if (!rc) {
        rc = tgmx_pipeline_step(&j.pipe, j.lo, j.n, j.call, &j.out, j.has_post ? &j.post : nullptr, (tgmx_stream_t)j.stream);
        if (rc) err = tgmx_last_error();
      }
      // the production event is recorded even after a failure: a consumer stream that waits on it must not wait for an older record
      if (j.record_ev && hipEventRecord(j.record_ev, j.stream) != hipSuccess && !rc) {
        rc = TGMX_E_LAUNCH;
        err = "worker: hipEventRecord failed";
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        if (rc) failed.emplace(j.ticket, std::make_pair(rc, err));
        done_ticket = j.ticket;
      }
      cv_done.notify_all();
    }
  }
};
}  // namespace

extern "C" int tgmx_worker_create(tgmx_worker_t* w) {
  TGMX_REQUIRE(w, "worker_create: null pointer");
  Worker* k = new Worker();
  if (hipGetDevice(&k->device) != hipSuccess) k->device = 0;  // the worker issues on the device current in the creating thread
  k->th = std::thread([k] { k->run(); });
  *w = (tgmx_worker_t)k;
  return TGMX_OK;
}

extern "C" int tgmx_worker_destroy(tgmx_worker_t w) {
  if (!w) return TGMX_OK;
  Worker* k = (Worker*)w;
  {
    std::lock_guard<std::mutex> lk(k->mu);
    k->stop = true;
  }
  k->cv_job.notify_all();
  if (k->th.joinable()) k->th.join();  // (pending jobs are finished first)
  delete k;
  return TGMX_OK;
}

extern "C" int tgmx_worker_pipeline_step(tgmx_worker_t w, const tgmx_pipeline_t* pipe, int64_t edge_lo, int64_t n_edges, uint64_t neg_call,
                                         const tgmx_pipeline_out_t* out, const tgmx_pipeline_post_t* post, tgmx_stream_t stream,
                                         tgmx_event_t wait_ev, tgmx_event_t record_ev, uint64_t* ticket) {
  TGMX_REQUIRE(w && pipe && out && ticket, "worker_pipeline_step: null pointer");
  Worker* k = (Worker*)w;
  Job j;
  j.pipe = *pipe;
  j.out = *out;
  j.has_post = post != nullptr;
  if (post) j.post = *post;
  j.lo = edge_lo;
  j.n = n_edges;
  j.call = neg_call;
  j.stream = (hipStream_t)stream;
  j.wait_ev = (hipEvent_t)wait_ev;
  j.record_ev = (hipEvent_t)record_ev;
  {
    std::lock_guard<std::mutex> lk(k->mu);
    j.ticket = k->next_ticket++;
    k->jobs.push_back(j);
  }
  k->cv_job.notify_one();
  *ticket = j.ticket;
  return TGMX_OK;
}

extern "C" int tgmx_worker_wait(tgmx_worker_t w, uint64_t ticket) {
  TGMX_REQUIRE(w, "worker_wait: null worker");
  Worker* k = (Worker*)w;
  std::unique_lock<std::mutex> lk(k->mu);
  TGMX_REQUIRE(ticket < k->next_ticket, "worker_wait: ticket %llu was never issued", (unsigned long long)ticket);
  k->cv_done.wait(lk, [&] { return k->done_ticket >= ticket; });
  auto it = k->failed.find(ticket);
  if (it == k->failed.end()) return TGMX_OK;
  const int rc = it->second.first;
  tgmx::set_error("%s", it->second.second.c_str());
  k->failed.erase(it);
  return rc;
}
