// Pipeline.h -- the per-GPU pipelined host path of the tools: a reader, one worker thread per GPU
// with its own libvc2hip context and pinned staging buffers, and an ordered writer.
//
// The reference's loop handles one picture at a time (EncodeStream.cpp:452-770, DecodeStream.cpp:289-613); pictures are
// independent, and only the ORDER of what is written chains (parse offsets, DataUnit.cpp:112-123; picture numbers,
// Utils.cpp:52-63).  So: picture k goes to worker k mod N; a worker keeps VC2HIP_MAX_INFLIGHT pictures in flight
// through the asynchronous picture calls of include/vc2hip.h (copy-in of one picture, kernels of another and copy-out
// of a third overlap); results come back to the caller in picture order.  Nothing here touches picture data on the
// CPU except the copies in and out of the staging buffers.
#ifndef VC2HOST_PIPELINE_H
#define VC2HOST_PIPELINE_H
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "vc2hip.h"

struct PictureResult {
  unsigned long long seq = 0;
  std::vector<unsigned char> bytes; // encode: the slice payload; decode: the raw picture
  std::vector<int> qidx;            // encode: the quantiser indices (ys * xs)
  std::string error;                // the exception text of a failed picture (thrown again by the caller, in order)
};

// File I/O at the speed of the link (round 3).  The round-2 tools moved every picture through ONE thread twice (istream
// read into the pinned buffer; a vector::assign copy of the result and an ostream write): 54 / 30 UHD frames per second
// where the GPU path sustains over a thousand.  Now
//   * a worker reads its own input: submitEncodeFile hands it (fd, offset, bytes) and the worker thread preads straight
//     into its pinned slot -- one reader per worker, in parallel;
//   * a worker delivers its own output: with a sink set, the finished picture is handed over as a pointer into the pinned
//     output buffer on the worker's thread (no copy); the sinks of the tools reserve the output range in picture order
//     (a short critical section) and pwrite outside it, so several workers write at once.
class GpuWorkers {
 public:
  // (seq, data, len, error): called on the worker thread that finished picture seq, pictures of one worker in order
  typedef std::function<void(unsigned long long, const unsigned char *, std::size_t, const std::string &)> Sink;
  void setSink(Sink s) { sink_ = s; }
  // encode picture seq from `bytes` bytes at `offset` of file descriptor fd (read by the worker into the slot reserved
  // with inputBuffer(seq))
  void submitEncodeFile(unsigned long long seq, int fd, long long offset, std::size_t bytes, const vc2hip_picture_format &pf,
                        const vc2hip_coding_params &cp, bool ld);
  // devices[g]: HIP device of worker g (a device may appear more than once: more pictures in flight on it).
  // in_bytes / out_bytes: upper bounds of one picture's input and output
  // qidx_ints: quantiser indices per picture that encode results carry (0: none wanted -- the stream writer does not use them)
  GpuWorkers(const std::vector<int> &devices, std::size_t in_bytes, std::size_t out_bytes, std::size_t qidx_ints = 0);
  ~GpuWorkers();
  // a pinned buffer of at least `bytes` (0: the in_bytes of the constructor) to put picture `seq`'s input in -- blocks
  // until the worker of seq has one free, and replaces it by a larger one when a picture needs more than the bound
  // given at construction (fragmented variable-size streams: the payload is the sum of its fragments) -- then submit it
  unsigned char *inputBuffer(unsigned long long seq, std::size_t bytes = 0);
  void submitEncode(unsigned long long seq, const vc2hip_picture_format &pf, const vc2hip_coding_params &cp, bool ld);
  void submitDecode(unsigned long long seq, std::size_t len, const vc2hip_picture_format &pf, const vc2hip_coding_params &cp, bool ld);
  bool poll(PictureResult &r); // the next picture in order, if it is done
  bool wait(PictureResult &r); // blocking; false once every submitted picture has been returned
  void close();                // no more submissions: the workers finish what they hold

 private:
  struct Job {
    unsigned long long seq;
    int slot;
    bool decode, ld;
    std::size_t len;
    vc2hip_picture_format pf;
    vc2hip_coding_params cp;
    int fd;            // >= 0: the worker reads the input itself
    long long offset;
  };
  struct Worker {
    int device = 0;
    std::thread th, reader;     // the GPU thread; the thread that reads file-sourced input (so that reads overlap the GPU thread's waits and writes)
    std::mutex m;
    std::condition_variable cv;
    std::deque<Job> queue;      // ready for the GPU thread
    std::deque<Job> toread;     // file-sourced jobs waiting for their bytes
    std::vector<unsigned char *> in, out; // pinned, SLOTS each
    std::vector<std::size_t> in_cap;      // bytes of in[i]
    std::vector<int *> qidx;              // pinned: the quantiser indices of an encoded picture come back here
    std::vector<bool> busy;               // input slot handed out and not yet finished
    int next_slot = 0;
    bool closing = false;
    bool reading = false;       // the reader holds a job (taken from toread, not yet in queue)
    std::string init_error;
  };
  static const int SLOTS = VC2HIP_MAX_INFLIGHT + 2; // two in flight, one being read / filled, one being written out
  void run(Worker &w);
  void runReader(Worker &w);
  void publish(PictureResult &&r);
  std::vector<Worker *> workers_;
  std::size_t in_bytes_, out_bytes_, qidx_ints_;
  Sink sink_;
  std::mutex rm_;
  std::condition_variable rcv_;
  std::map<unsigned long long, PictureResult> done_;
  std::map<unsigned long long, int> slot_of_; // seq -> input slot, between inputBuffer and submit
  unsigned long long next_out_ = 0, submitted_ = 0;
};
#endif
