// ptk_hostio.hpp -- the host-buffer entry of the k-NN search (ptk_search_knn): what a caller of the reference holds
// are numpy arrays / std::vectors in pageable host memory (the loop this replaces:
// /root/reference/src/pyco_tree/pico_tree/_pyco_tree/kd_tree.hpp:117-135), so this IS the path a drop-in user sees.
//
// A plain hipMemcpy of pageable memory goes through the runtime's single staging thread (6-27 GB/s on the boxes
// of round 2, 49 on one of round 3, for a link that does 48-56 GB/s from pinned memory: profiles/r03a_pcie.json), and
// upload, search and download would run one after the other.  Here the batch goes through in pieces:
//
//   caller's thread   piece i: copy the caller's rows into a pinned ring slot (CopyPool: a few host threads, 512 KB
//                     chunks; the copy of piece i + 1 is started before piece i is issued) -> async H2D -> search
//                     on one of two streams
//   download thread   piece i: once its search HAS finished, async D2H into a pinned ring slot -> copy into the
//                     caller's array.  (A D2H enqueued behind an event that has not fired yet sits at the head of the
//                     copy engine's queue and holds up the uploads of the following pieces behind it: measured, the
//                     pipeline then runs piece by piece.)
//
// so that the upload of piece i + 1, the search of piece i and the download of piece i - 1 overlap, in both
// directions of the link at once.  Rows come back in the caller's order whatever the pieces.

#pragma once

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace {

// A few threads that copy memory: copy() splits a block into chunks that the workers AND the caller take until none
// is left.  Several callers may be inside copy() at once (the two directions of a batch).
class CopyPool {
 public:
  explicit CopyPool(unsigned workers) {
    for (unsigned i = 0; i < workers; ++i) threads_.emplace_back([this] { work(); });
  }
  ~CopyPool() {
    {
      std::lock_guard<std::mutex> lock(mutex_);
      stop_ = true;
    }
    wake_.notify_all();
    for (std::thread& t : threads_) t.join();
  }
  CopyPool(const CopyPool&) = delete;
  CopyPool& operator=(const CopyPool&) = delete;

  struct Job;
  // Starts a copy on the workers and returns at once; finish() helps with what is left and waits for the rest.
  std::shared_ptr<Job> start(void* dst, const void* src, size_t bytes) {
    auto job = std::make_shared<Job>();
    job->dst = static_cast<char*>(dst);
    job->src = static_cast<const char*>(src);
    job->bytes = bytes;
    job->chunks = (bytes + kChunk - 1) / kChunk;
    if (job->chunks == 0) return job;
    if (threads_.empty()) {
      run(*job);
      return job;
    }
    {
      std::lock_guard<std::mutex> lock(mutex_);
      jobs_.push_back(job);
    }
    wake_.notify_all();
    return job;
  }
  void finish(const std::shared_ptr<Job>& job) {
    run(*job);
    std::unique_lock<std::mutex> lock(job->mutex);
    job->finished.wait(lock, [&] { return job->done.load() == job->chunks; });
  }
  void copy(void* dst, const void* src, size_t bytes) { finish(start(dst, src, bytes)); }

 private:
  static constexpr size_t kChunk = size_t(512) << 10;

 public:
  struct Job {
    char* dst;
    const char* src;
    size_t bytes, chunks;
    std::atomic<size_t> next{0}, done{0};
    std::mutex mutex;
    std::condition_variable finished;
  };

 private:
  // Takes chunks of the job until none is left.
  static void run(Job& job) {
    for (;;) {
      const size_t c = job.next.fetch_add(1);
      if (c >= job.chunks) return;
      const size_t lo = c * kChunk, n = std::min(kChunk, job.bytes - lo);
      std::memcpy(job.dst + lo, job.src + lo, n);
      if (job.done.fetch_add(1) + 1 == job.chunks) {
        std::lock_guard<std::mutex> lock(job.mutex);
        job.finished.notify_all();
      }
    }
  }
  void work() {
    for (;;) {
      std::shared_ptr<Job> job;
      {
        std::unique_lock<std::mutex> lock(mutex_);
        wake_.wait(lock, [&] { return stop_ || !jobs_.empty(); });
        if (stop_) return;
        job = jobs_.front();
        if (job->next.load() >= job->chunks) {  // handed out completely: not a job any more
          jobs_.pop_front();
          continue;
        }
      }
      run(*job);
    }
  }
  std::mutex mutex_;
  std::condition_variable wake_;
  std::deque<std::shared_ptr<Job>> jobs_;
  std::vector<std::thread> threads_;
  bool stop_ = false;
};

}  // namespace
