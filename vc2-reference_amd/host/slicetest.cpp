// slicetest -- needs a GPU.  The picture body of the reference's encoder and decoder written with the reference's own
// vocabulary (/root/reference/src/EncodeStream/EncodeStream.cpp:482-647, DecodeStream.cpp:451-613) on this host layer:
//   waveletTransform -> quantIndices -> quantise_transform_np -> split_into_blocks -> Slices ->
//   outStream << sliceio::highQualityVBR(prefix, scalar) << outSlices
// and back through operator>>, merge_blocks, inverse_quantise_transform_np, inverseWaveletTransform.  Checks that the
// stream idiom produces exactly the bytes of the direct C-ABI calls and that the round trip is the identity at q = 0.
#include <cstdio>
#include <sstream>
#include <string>

#include "Picture.h"
#include "Pipeline.h"
#include "Quantisation.h"
#include "Slices.h"
#include "WaveletTransform.h"

static int failures = 0;
#define EXPECT(cond) do { if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

static bool samePlanes(const Picture &a, const Picture &b) {
  const Array2D *pa[3] = {&a.y(), &a.c1(), &a.c2()}, *pb[3] = {&b.y(), &b.c1(), &b.c2()};
  for (int k = 0; k < 3; ++k) {
    if (pa[k]->num_elements() != pb[k]->num_elements()) return false;
    for (std::size_t i = 0; i < pa[k]->num_elements(); ++i) if (pa[k]->data()[i] != pb[k]->data()[i]) return false;
  }
  return true;
}

int main() {
  const int height = 64, width = 128, waveletDepth = 2, ySlices = 4, xSlices = 8, slicePrefix = 1, sliceScalar = 8;
  const WaveletKernel kernel = LeGall;
  const PictureFormat format(height, width, CF422);
  Picture picture(format);
  {
    Array2D y(format.lumaShape()), u(format.chromaShape()), v(format.chromaShape());
    unsigned seed = 12345;
    auto rnd = [&]() { seed = seed * 1103515245u + 12345u; return (int)((seed >> 16) & 1023) - 512; };
    for (std::size_t i = 0; i < y.num_elements(); ++i) y.data()[i] = rnd();
    for (std::size_t i = 0; i < u.num_elements(); ++i) { u.data()[i] = rnd(); v.data()[i] = rnd(); }
    picture.y(y); picture.c1(u); picture.c2(v);
  }
  const Array1D qMatrix = quantMatrix(kernel, waveletDepth);
  for (int qIndex : {0, 11}) {
    // ---- encoder body
    const Picture transform = waveletTransform(picture, kernel, waveletDepth);
    Array2D qIndices(ySlices, xSlices);
    for (std::size_t i = 0; i < qIndices.num_elements(); ++i) qIndices.data()[i] = qIndex; // quantIndicesConstQ
    const Picture quantisedSlices = quantise_transform_np(transform, qIndices, qMatrix);
    const PictureArray slices = split_into_blocks(quantisedSlices, ySlices, xSlices);
    const Slices outSlices(slices, waveletDepth, qIndices);
    std::ostringstream outStream;
    outStream << sliceio::highQualityVBR(slicePrefix, sliceScalar);
    outStream << outSlices;
    const std::string coded = outStream.str();
    const std::vector<unsigned char> direct = packSlicesHQ(quantisedSlices, waveletDepth, qIndices, slicePrefix, sliceScalar, nullptr);
    EXPECT(coded.size() == direct.size() && std::equal(direct.begin(), direct.end(), (const unsigned char *)coded.data()));
    // the first slice's length bytes are what component_slice_bytes says
    const Picture &s00 = slices[0][0];
    EXPECT((unsigned char)coded[slicePrefix] == qIndex);
    EXPECT((unsigned char)coded[slicePrefix + 1] * sliceScalar == component_slice_bytes(s00.y(), waveletDepth, sliceScalar));

    // ---- decoder body
    std::istringstream inStream(coded + "tail");
    Slices inSlices(quantisedSlices.format(), waveletDepth, ySlices, xSlices);
    inStream >> sliceio::highQualityVBR(slicePrefix, sliceScalar);
    inStream >> inSlices;
    EXPECT((std::size_t)inStream.tellg() == coded.size());           // exactly the slices were consumed
    const Picture yuvQCoeffs = merge_blocks(inSlices.yuvSlices);
    EXPECT(samePlanes(yuvQCoeffs, quantisedSlices));
    bool sameQ = true;
    for (std::size_t i = 0; i < qIndices.num_elements(); ++i) sameQ = sameQ && inSlices.qIndices.data()[i] == qIndex;
    EXPECT(sameQ);
    const Picture yuvTransform = inverse_quantise_transform_np(yuvQCoeffs, inSlices.qIndices, qMatrix);
    const Picture outPicture = inverseWaveletTransform(yuvTransform, kernel, waveletDepth, format);
    if (qIndex == 0) EXPECT(samePlanes(outPicture, picture));        // lossless at q = 0
  }
  { // HQ CBR through the same idiom: every slice is exactly its budget
    const Picture transform = waveletTransform(picture, kernel, waveletDepth);
    const Array2D sliceBytes = slice_bytes(ySlices, xSlices, 6000, sliceScalar);
    const Array2D qIndices = quantIndicesCBR(transform, qMatrix, sliceBytes, sliceScalar);
    const Picture quantisedSlices = quantise_transform_np(transform, qIndices, qMatrix);
    const Slices outSlices(split_into_blocks(quantisedSlices, ySlices, xSlices), waveletDepth, qIndices);
    std::ostringstream outStream;
    outStream << sliceio::highQualityCBR(sliceBytes, 0, sliceScalar) << outSlices;
    long want = 0;
    for (std::size_t i = 0; i < sliceBytes.num_elements(); ++i) want += sliceBytes.data()[i];
    EXPECT((long)outStream.str().size() == want);
  }
  { // The tools' worker pool (Pipeline.h) with pictures LARGER than the input bound it was built with -- what a fragmented
    // variable-size stream does to DecodeStream (the bound comes from the largest data unit, a picture is the sum of its
    // fragments): the slot's pinned buffer must grow, not overflow.  Eight pictures of growing payload through two workers
    // on device 0; the encoder side returns the quantiser indices through its pinned per-slot buffer.
    const vc2hip_picture_format pf = {width, height, (int)CF422, 10, 2, 0};
    const std::size_t rawBytes = vc2hip_raw_picture_bytes(&pf);
    std::vector<std::vector<unsigned char> > raws, payloads;
    std::vector<vc2hip_coding_params> cps;
    {
      GpuWorkers enc(std::vector<int>(2, 0), rawBytes, 200000, (std::size_t)ySlices * xSlices);
      for (int k = 0; k < 8; ++k) {
        std::vector<unsigned char> raw(rawBytes);
        unsigned seed = 99u + (unsigned)k;
        const int amp = 4 << k; // more detail, longer payload
        for (std::size_t i = 0; i + 1 < raw.size(); i += 2) {
          seed = seed * 1103515245u + 12345u;
          const unsigned v = (512u + (seed >> 16) % (unsigned)(amp > 1023 ? 1023 : amp)) & 1023u;
          raw[i] = (unsigned char)(v >> 2); raw[i + 1] = (unsigned char)((v & 3u) << 6); // 10 bits, MSB justified, big-endian
        }
        const vc2hip_coding_params cp = {(int)kernel, waveletDepth, ySlices, xSlices, VC2HIP_HQ_CONSTQ, k < 4 ? 3 : 0, 0, slicePrefix, sliceScalar};
        std::copy(raw.begin(), raw.end(), enc.inputBuffer((unsigned long long)k, rawBytes));
        enc.submitEncode((unsigned long long)k, pf, cp, false);
        raws.push_back(raw); cps.push_back(cp);
      }
      enc.close();
      PictureResult r;
      while (enc.wait(r)) {
        EXPECT(r.error.empty());
        EXPECT(r.qidx.size() == (std::size_t)ySlices * xSlices && r.qidx[0] == cps[(std::size_t)r.seq].q_index);
        payloads.push_back(r.bytes);
      }
    }
    EXPECT(payloads.size() == 8 && payloads[7].size() > payloads[0].size() + 64);
    GpuWorkers dec(std::vector<int>(2, 0), payloads.empty() ? 16 : payloads[0].size(), rawBytes); // bound = the FIRST (smallest) picture
    for (std::size_t k = 0; k < payloads.size(); ++k) {
      std::copy(payloads[k].begin(), payloads[k].end(), dec.inputBuffer(k, payloads[k].size()));
      dec.submitDecode(k, payloads[k].size(), pf, cps[k], false);
    }
    dec.close();
    PictureResult r;
    std::size_t got = 0;
    while (dec.wait(r)) {
      EXPECT(r.error.empty() && r.bytes.size() == rawBytes);
      if (cps[(std::size_t)r.seq].q_index == 0) EXPECT(r.bytes == raws[(std::size_t)r.seq]); // index 0: lossless
      ++got;
    }
    EXPECT(got == payloads.size());
  }
  if (failures) { std::printf("%d failure(s)\n", failures); return 1; }
  std::printf("slicetest ok\n");
  return 0;
}
