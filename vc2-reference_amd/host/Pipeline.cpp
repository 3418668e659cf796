#include "Pipeline.h"

#include <cerrno>
#include <cstring>
#include <stdexcept>

#include <unistd.h>

#include "Hip.h"

GpuWorkers::GpuWorkers(const std::vector<int> &devices, std::size_t in_bytes, std::size_t out_bytes, std::size_t qidx_ints)
    : in_bytes_(in_bytes), out_bytes_(out_bytes), qidx_ints_(qidx_ints) {
  for (int d : devices) (void)hipContext(d); // fails loudly here when a device is missing: there is no CPU fallback
  for (int d : devices) {
    Worker *w = new Worker;
    w->device = d;
    w->busy.assign(SLOTS, false);
    for (int i = 0; i < SLOTS; ++i) {
      w->in.push_back((unsigned char *)vc2hip_host_alloc(in_bytes + 64));
      w->in_cap.push_back(in_bytes + 64);
      w->out.push_back((unsigned char *)vc2hip_host_alloc(out_bytes + 64));
      w->qidx.push_back(qidx_ints ? (int *)vc2hip_host_alloc(qidx_ints * sizeof(int)) : nullptr);
      if (!w->in.back() || !w->out.back() || (qidx_ints && !w->qidx.back())) throw std::runtime_error("vc2hip: cannot allocate pinned staging buffers");
    }
    workers_.push_back(w);
  }
  for (Worker *w : workers_) {
    w->th = std::thread([this, w]() { run(*w); });
    w->reader = std::thread([this, w]() { runReader(*w); });
  }
}

GpuWorkers::~GpuWorkers() {
  close();
  for (Worker *w : workers_) {
    if (w->reader.joinable()) w->reader.join();
    if (w->th.joinable()) w->th.join();
    for (unsigned char *p : w->in) vc2hip_host_free(p);
    for (unsigned char *p : w->out) vc2hip_host_free(p);
    for (int *p : w->qidx) vc2hip_host_free(p);
    delete w;
  }
}

unsigned char *GpuWorkers::inputBuffer(unsigned long long seq, std::size_t bytes) {
  Worker &w = *workers_[(std::size_t)(seq % workers_.size())];
  std::unique_lock<std::mutex> lock(w.m);
  w.cv.wait(lock, [&]() { return !w.busy[(std::size_t)w.next_slot]; });
  const int slot = w.next_slot;
  w.busy[(std::size_t)slot] = true;
  w.next_slot = (slot + 1) % SLOTS;
  lock.unlock();
  if (bytes + 64 > w.in_cap[(std::size_t)slot]) { // the slot is free (nothing in flight reads it): a larger buffer takes its place
    unsigned char *bigger = (unsigned char *)vc2hip_host_alloc(bytes + bytes / 4 + 64);
    if (!bigger) throw std::runtime_error("vc2hip: cannot allocate a pinned staging buffer for a picture of " + std::to_string(bytes) + " bytes");
    vc2hip_host_free(w.in[(std::size_t)slot]);
    w.in[(std::size_t)slot] = bigger;
    w.in_cap[(std::size_t)slot] = bytes + bytes / 4 + 64;
  }
  std::lock_guard<std::mutex> rl(rm_);
  slot_of_[seq] = slot;
  return w.in[(std::size_t)slot];
}

void GpuWorkers::submitEncode(unsigned long long seq, const vc2hip_picture_format &pf, const vc2hip_coding_params &cp, bool ld) {
  Job j = {seq, 0, false, ld, 0, pf, cp, -1, 0};
  { std::lock_guard<std::mutex> rl(rm_); j.slot = slot_of_[seq]; slot_of_.erase(seq); ++submitted_; }
  Worker &w = *workers_[(std::size_t)(seq % workers_.size())];
  { std::lock_guard<std::mutex> lock(w.m); w.queue.push_back(j); }
  w.cv.notify_all();
}

void GpuWorkers::submitEncodeFile(unsigned long long seq, int fd, long long offset, std::size_t bytes, const vc2hip_picture_format &pf,
                                  const vc2hip_coding_params &cp, bool ld) {
  Job j = {seq, 0, false, ld, bytes, pf, cp, fd, offset};
  { std::lock_guard<std::mutex> rl(rm_); j.slot = slot_of_[seq]; slot_of_.erase(seq); ++submitted_; }
  Worker &w = *workers_[(std::size_t)(seq % workers_.size())];
  { std::lock_guard<std::mutex> lock(w.m); w.toread.push_back(j); }
  w.cv.notify_all();
}

// a worker's reader: file-sourced pictures, in order, straight into the pinned slot; then over to the GPU thread
void GpuWorkers::runReader(Worker &w) {
  for (;;) {
    Job j;
    {
      std::unique_lock<std::mutex> lock(w.m);
      w.cv.wait(lock, [&]() { return !w.toread.empty() || w.closing; });
      if (w.toread.empty()) return;
      j = w.toread.front();
      w.toread.pop_front();
      w.reading = true;
    }
    std::size_t got = 0;
    while (got < j.len) {
      const ssize_t r = pread(j.fd, w.in[(std::size_t)j.slot] + got, j.len - got, (off_t)(j.offset + (long long)got));
      if (r < 0 && errno == EINTR) continue;
      if (r <= 0) break;
      got += (std::size_t)r;
    }
    j.fd = got == j.len ? -1 : -2; // -2: the GPU thread reports the short read, in order
    { std::lock_guard<std::mutex> lock(w.m); w.queue.push_back(j); w.reading = false; }
    w.cv.notify_all();
  }
}

void GpuWorkers::submitDecode(unsigned long long seq, std::size_t len, const vc2hip_picture_format &pf, const vc2hip_coding_params &cp, bool ld) {
  Job j = {seq, 0, true, ld, len, pf, cp, -1, 0};
  { std::lock_guard<std::mutex> rl(rm_); j.slot = slot_of_[seq]; slot_of_.erase(seq); ++submitted_; }
  Worker &w = *workers_[(std::size_t)(seq % workers_.size())];
  { std::lock_guard<std::mutex> lock(w.m); w.queue.push_back(j); }
  w.cv.notify_all();
}

void GpuWorkers::close() {
  for (Worker *w : workers_) {
    { std::lock_guard<std::mutex> lock(w->m); w->closing = true; }
    w->cv.notify_all();
  }
}

void GpuWorkers::publish(PictureResult &&r) {
  { std::lock_guard<std::mutex> lock(rm_); done_[r.seq] = std::move(r); }
  rcv_.notify_all();
}

bool GpuWorkers::poll(PictureResult &r) {
  std::lock_guard<std::mutex> lock(rm_);
  auto it = done_.find(next_out_);
  if (it == done_.end()) return false;
  r = std::move(it->second);
  done_.erase(it);
  ++next_out_;
  return true;
}

bool GpuWorkers::wait(PictureResult &r) {
  std::unique_lock<std::mutex> lock(rm_);
  if (next_out_ >= submitted_) return false;
  rcv_.wait(lock, [&]() { return done_.count(next_out_) != 0; });
  auto it = done_.find(next_out_);
  r = std::move(it->second);
  done_.erase(it);
  ++next_out_;
  return true;
}

// One worker: its own context on its device; up to VC2HIP_MAX_INFLIGHT pictures begun and not yet ended.  A new
// picture is begun as soon as it arrives; the oldest one is ended when the flight is full or nothing else is waiting.
void GpuWorkers::run(Worker &w) {
  vc2hip_ctx *ctx = nullptr;
  const int rc = vc2hip_create(w.device, &ctx);
  if (rc != VC2HIP_OK)
    w.init_error = std::string("vc2hip: cannot create a context on HIP device ") + std::to_string(w.device) + " (" +
                   vc2hip_error_string(rc) + "); there is no CPU fallback";
  struct Open { Job job; int ticket; PictureResult res; };
  std::deque<Open> open;
  auto finish = [&](Open &o) {
    std::size_t len = 0;
    if (o.res.error.empty()) {
      if (o.job.decode) {
        const int e = vc2hip_decode_picture_end(ctx, o.ticket);
        if (e) o.res.error = vc2hip_last_error(ctx);
        else len = vc2hip_raw_picture_bytes(&o.job.pf);
      } else {
        const int e = vc2hip_encode_picture_end(ctx, o.ticket, &len);
        if (e) o.res.error = vc2hip_last_error(ctx);
        else if (qidx_ints_) o.res.qidx.assign(w.qidx[(std::size_t)o.job.slot], w.qidx[(std::size_t)o.job.slot] + (std::size_t)o.job.cp.y_slices * o.job.cp.x_slices);
      }
    }
    const unsigned char *data = w.out[(std::size_t)o.job.slot];
    if (sink_) { // straight from the pinned buffer, on this thread
      try { sink_(o.job.seq, data, o.res.error.empty() ? len : 0, o.res.error); }
      catch (const std::exception &ex) { if (o.res.error.empty()) o.res.error = ex.what(); }
    } else if (o.res.error.empty()) o.res.bytes.assign(data, data + len);
    { std::lock_guard<std::mutex> lock(w.m); w.busy[(std::size_t)o.job.slot] = false; }
    w.cv.notify_all();
    publish(std::move(o.res));
  };
  for (;;) {
    Job j;
    bool have = false;
    {
      std::unique_lock<std::mutex> lock(w.m);
      if (open.empty()) w.cv.wait(lock, [&]() { return !w.queue.empty() || (w.closing && w.toread.empty() && !w.reading); });
      if (!w.queue.empty()) { j = w.queue.front(); w.queue.pop_front(); have = true; }
      else if (open.empty() && w.closing && w.toread.empty() && !w.reading) break;
    }
    if (!have) { finish(open.front()); open.pop_front(); continue; } // nothing waiting: hand the oldest picture back
    if ((int)open.size() == VC2HIP_MAX_INFLIGHT) { finish(open.front()); open.pop_front(); }
    Open o;
    o.job = j;
    o.ticket = -1;
    o.res.seq = j.seq;
    if (!w.init_error.empty()) o.res.error = w.init_error;
    else if (j.fd == -2) o.res.error = "vc2hip: short read of an input picture";
    if (!o.res.error.empty()) {}
    else if (j.decode) {
      const int e = vc2hip_decode_picture_begin(ctx, w.in[(std::size_t)j.slot], j.len, &j.pf, &j.cp, w.out[(std::size_t)j.slot], &o.ticket);
      if (e) o.res.error = vc2hip_last_error(ctx);
    } else {
      // (the indices, when wanted, land in the slot's pinned buffer: a pageable destination would make _begin block until the
      // picture's kernels have run -- no second picture in flight)
      int *qdst = qidx_ints_ >= (std::size_t)j.cp.y_slices * j.cp.x_slices ? w.qidx[(std::size_t)j.slot] : nullptr;
      const int e = vc2hip_encode_picture_begin(ctx, w.in[(std::size_t)j.slot], &j.pf, &j.cp, w.out[(std::size_t)j.slot], out_bytes_,
                                                qdst, &o.ticket);
      if (e) o.res.error = vc2hip_last_error(ctx);
    }
    open.push_back(std::move(o));
  }
  if (ctx) vc2hip_destroy(ctx);
}
