// Arrays.h -- Boost-free containers with the subset of the reference's Arrays.h API that the hot
// path and the tools use (/root/reference/src/Library/Arrays.h:17-50, src/Arrays.cpp:41-111):
// contiguous row-major int planes, blocks of planes, clip, split/merge into slices.
#ifndef VC2HOST_ARRAYS_H
#define VC2HOST_ARRAYS_H

#include <cstddef>
#include <vector>

typedef std::ptrdiff_t Index;

struct Shape2D {
  Index d[2];
  Index &operator[](int i) { return d[i]; }
  const Index &operator[](int i) const { return d[i]; }
};

class Array1D {
 public:
  explicit Array1D(std::size_t n = 0) : v_(n, 0) {}
  int &operator[](std::size_t i) { return v_[i]; }
  const int &operator[](std::size_t i) const { return v_[i]; }
  std::size_t size() const { return v_.size(); }
  int *data() { return v_.data(); }
  const int *data() const { return v_.data(); }

 private:
  std::vector<int> v_;
};

class Array2D {
 public:
  Array2D() : h_(0), w_(0) {}
  Array2D(Index h, Index w) : h_(h), w_(w), v_((std::size_t)(h * w), 0) {}
  explicit Array2D(const Shape2D &s) : h_(s[0]), w_(s[1]), v_((std::size_t)(s[0] * s[1]), 0) {}
  int *operator[](Index y) { return v_.data() + y * w_; }
  const int *operator[](Index y) const { return v_.data() + y * w_; }
  Shape2D shape() const { Shape2D s = {{h_, w_}}; return s; }
  std::size_t num_elements() const { return v_.size(); }
  int *data() { return v_.data(); }
  const int *data() const { return v_.data(); }
  // multi_array::resize: keeps the overlapping top-left, zero-fills the rest
  void resize(const Shape2D &s);

 private:
  Index h_, w_;
  std::vector<int> v_;
};

// Range(start, finish, stride) / Range(): the index ranges the reference writes as indices[Range(..)][Range(..)]
// (Arrays.h:17-50); a default Range is the whole extent
struct Range {
  Index start, finish, stride;
  bool all;
  Range() : start(0), finish(0), stride(1), all(true) {}
  Range(Index s, Index f, Index st = 1) : start(s), finish(f), stride(st), all(false) {}
  Index first(Index) const { return all ? 0 : start; }
  Index count(Index extent) const { return all ? extent : (finish > start ? (finish - start + stride - 1) / stride : 0); }
};

// a strided window onto an Array2D (the reference's View2D / ConstView2D): reads, writes, element-wise copies
class View2D {
 public:
  View2D(Array2D &a, const Range &rows, const Range &cols)
      : base_(a.data() + rows.first(a.shape()[0]) * a.shape()[1] + cols.first(a.shape()[1])),
        rs_(rows.stride * a.shape()[1]), cs_(cols.stride), h_(rows.count(a.shape()[0])), w_(cols.count(a.shape()[1])) {}
  View2D(const Array2D &a, const Range &rows, const Range &cols) : View2D(const_cast<Array2D &>(a), rows, cols) {}
  Shape2D shape() const { Shape2D s = {{h_, w_}}; return s; }
  int &at(Index y, Index x) { return base_[y * rs_ + x * cs_]; }
  int at(Index y, Index x) const { return base_[y * rs_ + x * cs_]; }
  View2D &operator=(const Array2D &src) { // element copy; shapes must agree
    for (Index y = 0; y < h_; ++y) for (Index x = 0; x < w_; ++x) at(y, x) = src[y][x];
    return *this;
  }
  operator Array2D() const { // a dense copy (what assigning a view to an Array2D does in the reference)
    Array2D out(h_, w_);
    for (Index y = 0; y < h_; ++y) for (Index x = 0; x < w_; ++x) out[y][x] = at(y, x);
    return out;
  }

 private:
  int *base_;
  Index rs_, cs_, h_, w_;
};

typedef std::vector<Array2D> BlockVector; // the subbands of a transform, LL first (Arrays.h)

// ySlices x xSlices blocks of planes
struct BlockArray {
  int ys, xs;
  std::vector<Array2D> blocks;
  BlockArray(int y = 0, int x = 0) : ys(y), xs(x), blocks((std::size_t)(y * x)) {}
  Array2D &at(int v, int h) { return blocks[(std::size_t)(v * xs + h)]; }
  const Array2D &at(int v, int h) const { return blocks[(std::size_t)(v * xs + h)]; }
};

const Array2D clip(const Array2D &values, int min_value, int max_value);          // Arrays.cpp:41-53
const BlockArray split_into_blocks(const Array2D &picture, int yBlocks, int xBlocks); // Arrays.cpp:60-78
const Array2D merge_blocks(const BlockArray &blocks);                              // Arrays.cpp:83-111

#endif
