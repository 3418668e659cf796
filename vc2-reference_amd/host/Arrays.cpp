#include "Arrays.h"
#include "WaveletTransform.h"

#include <algorithm>
#include <cstring>

void Array2D::resize(const Shape2D &s) {
  std::vector<int> nv((std::size_t)(s[0] * s[1]), 0);
  const Index ch = std::min(h_, s[0]), cw = std::min(w_, s[1]);
  for (Index y = 0; y < ch; ++y) std::memcpy(nv.data() + y * s[1], v_.data() + y * w_, (std::size_t)cw * sizeof(int));
  v_.swap(nv);
  h_ = s[0];
  w_ = s[1];
}

const Array2D clip(const Array2D &values, int min_value, int max_value) {
  Array2D out(values.shape());
  for (std::size_t i = 0; i < values.num_elements(); ++i)
    out.data()[i] = std::min(std::max(values.data()[i], min_value), max_value);
  return out;
}

const BlockArray split_into_blocks(const Array2D &picture, int yBlocks, int xBlocks) {
  const Index bh = picture.shape()[0] / yBlocks, bw = picture.shape()[1] / xBlocks;
  BlockArray out(yBlocks, xBlocks);
  for (int v = 0; v < yBlocks; ++v)
    for (int h = 0; h < xBlocks; ++h) {
      Array2D b(bh, bw);
      for (Index y = 0; y < bh; ++y) std::memcpy(b[y], picture[v * bh + y] + h * bw, (std::size_t)bw * sizeof(int));
      out.at(v, h) = b;
    }
  return out;
}

const Array2D merge_blocks(const BlockArray &blocks) {
  const Index bh = blocks.at(0, 0).shape()[0], bw = blocks.at(0, 0).shape()[1];
  Array2D out(bh * blocks.ys, bw * blocks.xs);
  for (int v = 0; v < blocks.ys; ++v)
    for (int h = 0; h < blocks.xs; ++h)
      for (Index y = 0; y < bh; ++y)
        std::memcpy(out[v * bh + y] + h * bw, blocks.at(v, h)[y], (std::size_t)bw * sizeof(int));
  return out;
}

// WaveletTransform.cpp:428-450: subband b of an in-place transform as a dense array.  LL: stride 2^depth, phase 0;
// level L (1 .. depth): stride s = 2^(depth + 1 - L), phase s / 2 -- HL (columns), LH (rows), HH (both)
const BlockVector split_into_subbands(const Array2D &picture, char waveletDepth) {
  const Index H = picture.shape()[0], W = picture.shape()[1];
  BlockVector bands;
  Index stride = (Index)1 << waveletDepth;
  bands.push_back(View2D(picture, Range(0, H, stride), Range(0, W, stride)));
  for (char level = 1; level <= waveletDepth; ++level) {
    stride = (Index)1 << (waveletDepth + 1 - level);
    const Index offset = stride / 2;
    bands.push_back(View2D(picture, Range(0, H, stride), Range(offset, W, stride)));
    bands.push_back(View2D(picture, Range(offset, H, stride), Range(0, W, stride)));
    bands.push_back(View2D(picture, Range(offset, H, stride), Range(offset, W, stride)));
  }
  return bands;
}

// WaveletTransform.cpp:454-476
const Array2D merge_subbands(const BlockVector &subbands) {
  const char waveletDepth = (char)((subbands.size() - 1) / 3);
  const Index H = subbands[0].shape()[0] << waveletDepth, W = subbands[0].shape()[1] << waveletDepth;
  Array2D picture(H, W);
  Index stride = (Index)1 << waveletDepth;
  View2D(picture, Range(0, H, stride), Range(0, W, stride)) = subbands[0];
  std::size_t band = 1;
  for (char level = 1; level <= waveletDepth; ++level) {
    stride = (Index)1 << (waveletDepth + 1 - level);
    const Index offset = stride / 2;
    View2D(picture, Range(0, H, stride), Range(offset, W, stride)) = subbands[band++];
    View2D(picture, Range(offset, H, stride), Range(0, W, stride)) = subbands[band++];
    View2D(picture, Range(offset, H, stride), Range(offset, W, stride)) = subbands[band++];
  }
  return picture;
}
