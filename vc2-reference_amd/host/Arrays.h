// Arrays.h -- Boost-free containers with the subset of the reference's Arrays.h API that the hot
// path and the tools use (/root/reference/src/Library/Arrays.h:17-50, src/Arrays.cpp:41-111):
// contiguous row-major int planes, blocks of planes, clip, split/merge into slices -- with the
// boost::multi_array idioms the reference's sources are written in, so that its bodies compile here as they stand:
//   Array2D a(extents[h][w]);  a.resize(extents[h][w]);  Array2D b(a.ranges());  Array1D m(extents[n]);
//   a[indices[Range(top, bottom, 2)][Range()]] as an r-value (a dense copy) and as an l-value (element-wise assignment)
//   view[y][x],  BlockArray blocks(extents[ys][xs]);  blocks[v][h],  BlockVector bands(extents[n])
// (EncodeStream.cpp:80,133,161-162,201; Frame.cpp:52-94; Arrays.cpp:42,62,92; WaveletTransform.cpp:349,433,460)
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

// boost::extents: extents[a] and extents[a][b]
struct Extents2 { Index d[2]; };
struct Extents1 { Index d0; Extents2 operator[](Index b) const { Extents2 e = {{d0, b}}; return e; } };
struct ExtentGen { Extents1 operator[](Index a) const { Extents1 e = {a}; return e; } };
static const ExtentGen extents = ExtentGen();

class Array1D {
 public:
  explicit Array1D(std::size_t n = 0) : v_(n, 0) {}
  Array1D(const Extents1 &e) : v_((std::size_t)e.d0, 0) {}
  Extents1 ranges() const { Extents1 e = {(Index)v_.size()}; return e; }
  void resize(const Extents1 &e) { v_.resize((std::size_t)e.d0, 0); }
  int &operator[](std::size_t i) { return v_[i]; }
  const int &operator[](std::size_t i) const { return v_[i]; }
  std::size_t size() const { return v_.size(); }
  int *data() { return v_.data(); }
  const int *data() const { return v_.data(); }

 private:
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

// boost::indices: indices[Range][Range] (the reference's ArrayIndices2D)
struct ArrayIndices2D { Range r[2]; };
struct Indices1 { Range r0; ArrayIndices2D operator[](const Range &b) const { ArrayIndices2D i; i.r[0] = r0; i.r[1] = b; return i; } };
struct IndexGen { Indices1 operator[](const Range &a) const { Indices1 i; i.r0 = a; return i; } };
static const IndexGen indices = IndexGen();
// The reference's Arrays.h:17-18 imports both generators from namespace boost, and some of its sources say
// `using boost::indices;` again inside a function (Frame.cpp:51).  The qualified names resolve to the objects above,
// so such call sites build against this header unchanged.  (Names only: there is no Boost here and none is imitated.)
namespace boost { using ::extents; using ::indices; }

class View2D;
class Array2D {
 public:
  Array2D() : h_(0), w_(0) {}
  Array2D(const Extents2 &e) : h_(e.d[0]), w_(e.d[1]), v_((std::size_t)(e.d[0] * e.d[1]), 0) {}
  Extents2 ranges() const { Extents2 e = {{h_, w_}}; return e; }
  void resize(const Extents2 &e) { Shape2D s = {{e.d[0], e.d[1]}}; resize(s); }
  inline View2D operator[](const ArrayIndices2D &ix);             // a[indices[..][..]]: assignable window
  inline const View2D operator[](const ArrayIndices2D &ix) const; //   ... of a const array: read, or copy into an Array2D
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

// a strided window onto an Array2D (the reference's View2D / ConstView2D): reads, writes, element-wise copies
class View2D {
 public:
  View2D(Array2D &a, const Range &rows, const Range &cols)
      : base_(a.data() + rows.first(a.shape()[0]) * a.shape()[1] + cols.first(a.shape()[1])),
        rs_(rows.stride * a.shape()[1]), cs_(cols.stride), h_(rows.count(a.shape()[0])), w_(cols.count(a.shape()[1])) {}
  View2D(const Array2D &a, const Range &rows, const Range &cols) : View2D(const_cast<Array2D &>(a), rows, cols) {}
  Shape2D shape() const { Shape2D s = {{h_, w_}}; return s; }
  struct Row { // view[y][x]
    int *p; Index cs;
    int &operator[](Index x) const { return p[x * cs]; }
  };
  Row operator[](Index y) { Row r = {base_ + y * rs_, cs_}; return r; }
  const Row operator[](Index y) const { Row r = {base_ + y * rs_, cs_}; return r; }
  View2D &operator=(const View2D &src) { // view = view: element copy (not a rebinding), as in multi_array
    for (Index y = 0; y < h_; ++y) for (Index x = 0; x < w_; ++x) at(y, x) = src.at(y, x);
    return *this;
  }
  View2D(const View2D &) = default;
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

typedef View2D ConstView2D;
inline View2D Array2D::operator[](const ArrayIndices2D &ix) { return View2D(*this, ix.r[0], ix.r[1]); }
inline const View2D Array2D::operator[](const ArrayIndices2D &ix) const { return View2D(*this, ix.r[0], ix.r[1]); }
inline const Shape2D shape(const Array2D &a) { return a.shape(); }
inline const Shape2D shape(const View2D &v) { return v.shape(); }

// the subbands of a transform, LL first (Arrays.h): a vector of planes, constructible from extents[n]
class BlockVector : public std::vector<Array2D> {
 public:
  BlockVector() {}
  BlockVector(const Extents1 &e) : std::vector<Array2D>((std::size_t)e.d0) {}
  Extents1 ranges() const { Extents1 e = {(Index)size()}; return e; }
};

// ySlices x xSlices blocks of planes
struct BlockArray {
  int ys, xs;
  std::vector<Array2D> blocks;
  BlockArray(int y = 0, int x = 0) : ys(y), xs(x), blocks((std::size_t)(y * x)) {}
  BlockArray(const Extents2 &e) : ys((int)e.d[0]), xs((int)e.d[1]), blocks((std::size_t)(e.d[0] * e.d[1])) {}
  Array2D *operator[](Index v) { return &blocks[(std::size_t)(v * xs)]; }             // blocks[v][h]
  const Array2D *operator[](Index v) const { return &blocks[(std::size_t)(v * xs)]; }
  Shape2D shape() const { Shape2D s = {{ys, xs}}; return s; }
  Array2D &at(int v, int h) { return blocks[(std::size_t)(v * xs + h)]; }
  const Array2D &at(int v, int h) const { return blocks[(std::size_t)(v * xs + h)]; }
};

inline const Shape2D shape(const BlockArray &b) { return b.shape(); }
const Array2D clip(const Array2D &values, int min_value, int max_value);          // Arrays.cpp:41-53
const BlockArray split_into_blocks(const Array2D &picture, int yBlocks, int xBlocks); // Arrays.cpp:60-78
const Array2D merge_blocks(const BlockArray &blocks);                              // Arrays.cpp:83-111

#endif
