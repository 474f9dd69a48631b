// BC7 block decode (D3D11 functional specification, "BC7 format"): 16 bytes -> 16 RGBA8 texels.
//
// The reference hands BC7 data (DDS via nv_dds, KTX/KTX2 via nv_ktx: src/gltf_image_loader.cpp:69-160) to the GPU's texture unit;
// this path has none, so the block is decoded on the host.  Eight modes selected by the position of the first set bit: 1-3 subsets
// chosen by a partition shape, endpoints of 4-8 bits per channel optionally extended by a parity bit, 2-4 bit indices into a
// 64-step interpolation, the anchor texel of every subset stored without its top index bit; modes 4/5 carry separate colour and
// alpha indices and a channel rotation.  Partition / anchor tables: bc7_tables.inc (generated and verified by
// tools/gen_bc7_tables.py); tests/test_host_loader.py::test_bc7_decode compares random blocks of every mode with Pillow's decoder.
#include "image_loader.hpp"

namespace mihost {

namespace {

#include "bc7_tables.inc"

struct BitReader
{
  const uint8_t* s;
  int            pos = 0;
  uint32_t       get(int bits)
  {
    uint32_t v = 0;
    for(int i = 0; i < bits; ++i, ++pos)
      v |= uint32_t((s[pos >> 3] >> (pos & 7)) & 1u) << i;
    return v;
  }
};

struct ModeInfo
{
  int subsets, partitionBits, rotationBits, indexSelectionBit, colorBits, alphaBits, endpointPBits, sharedPBits, indexBits, index2Bits;
};
const ModeInfo kModes[8] = {
    {3, 4, 0, 0, 4, 0, 1, 0, 3, 0}, {2, 6, 0, 0, 6, 0, 0, 1, 3, 0}, {3, 6, 0, 0, 5, 0, 0, 0, 2, 0}, {2, 6, 0, 0, 7, 0, 1, 0, 2, 0},
    {1, 0, 2, 1, 5, 6, 0, 0, 2, 3}, {1, 0, 2, 0, 7, 8, 0, 0, 2, 2}, {1, 0, 0, 0, 7, 7, 1, 0, 4, 0}, {2, 6, 0, 0, 5, 5, 1, 0, 2, 0},
};
const int kWeights2[4]  = {0, 21, 43, 64};
const int kWeights3[8]  = {0, 9, 18, 27, 37, 46, 55, 64};
const int kWeights4[16] = {0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64};

int interpolate(int e0, int e1, int index, int bits)
{
  const int w = bits == 2 ? kWeights2[index] : bits == 3 ? kWeights3[index] : kWeights4[index];
  return ((64 - w) * e0 + w * e1 + 32) >> 6;
}

}  // namespace

void decodeBc7Block(const uint8_t* block, uint8_t out[16][4])
{
  int mode = 0;
  while(mode < 8 && !((block[0] >> mode) & 1))
    ++mode;
  if(mode == 8)  // reserved: the specification returns zeros
  {
    for(int i = 0; i < 16; ++i)
      out[i][0] = out[i][1] = out[i][2] = out[i][3] = 0;
    return;
  }
  const ModeInfo& m = kModes[mode];
  BitReader       br{block, mode + 1};
  const int       partition = int(br.get(m.partitionBits));
  const int       rotation  = int(br.get(m.rotationBits));
  const int       indexSel  = int(br.get(m.indexSelectionBit));
  const int       numEnd    = m.subsets * 2;
  int             e[6][4];
  for(int c = 0; c < 3; ++c)
    for(int k = 0; k < numEnd; ++k)
      e[k][c] = int(br.get(m.colorBits));
  for(int k = 0; k < numEnd; ++k)
    e[k][3] = m.alphaBits ? int(br.get(m.alphaBits)) : 255;
  // parity bits: one per endpoint, or one shared by the two endpoints of a subset
  int colorBits = m.colorBits, alphaBits = m.alphaBits;
  if(m.endpointPBits || m.sharedPBits)
  {
    int shared = 0;  // a shared parity bit is read at the even endpoint and applies to both endpoints of the subset
    for(int k = 0; k < numEnd; ++k)
    {
      if(m.endpointPBits || (k & 1) == 0)
        shared = int(br.get(1));
      const int p = shared;
      for(int c = 0; c < 3; ++c)
        e[k][c] = (e[k][c] << 1) | p;
      if(m.alphaBits)
        e[k][3] = (e[k][3] << 1) | p;
    }
    ++colorBits;
    if(m.alphaBits)
      ++alphaBits;
  }
  // expand to 8 bits by replicating the top bits
  for(int k = 0; k < numEnd; ++k)
  {
    for(int c = 0; c < 3; ++c)
    {
      const int v = e[k][c] << (8 - colorBits);
      e[k][c]     = v | (v >> colorBits);
    }
    if(m.alphaBits)
    {
      const int v = e[k][3] << (8 - alphaBits);
      e[k][3]     = v | (v >> alphaBits);
    }
  }
  // indices: the anchor texel of each subset has one bit less
  const uint8_t* shape = m.subsets == 2 ? kBc7Partition2[partition] : (m.subsets == 3 ? kBc7Partition3[partition] : nullptr);
  int            anchor[3] = {0, 0, 0};
  if(m.subsets == 2)
    anchor[1] = kBc7Anchor2[partition];
  else if(m.subsets == 3)
  {
    anchor[1] = kBc7Anchor3a[partition];
    anchor[2] = kBc7Anchor3b[partition];
  }
  int idx1[16], idx2[16];
  for(int i = 0; i < 16; ++i)
  {
    const int s = shape ? shape[i] : 0;
    idx1[i]     = int(br.get(i == anchor[s] ? m.indexBits - 1 : m.indexBits));
  }
  for(int i = 0; i < 16; ++i)
    idx2[i] = m.index2Bits ? int(br.get(i == 0 ? m.index2Bits - 1 : m.index2Bits)) : 0;
  for(int i = 0; i < 16; ++i)
  {
    const int  s  = shape ? shape[i] : 0;
    const int *e0 = e[2 * s], *e1 = e[2 * s + 1];
    int        rgba[4];
    if(m.index2Bits == 0)
    {
      for(int c = 0; c < 4; ++c)
        rgba[c] = interpolate(e0[c], e1[c], idx1[i], m.indexBits);
    }
    else
    {
      // modes 4 / 5: colour and alpha have their own index sets; the index-selection bit of mode 4 swaps which set drives which
      const int cIdx = indexSel ? idx2[i] : idx1[i], cBits = indexSel ? m.index2Bits : m.indexBits;
      const int aIdx = indexSel ? idx1[i] : idx2[i], aBits = indexSel ? m.indexBits : m.index2Bits;
      for(int c = 0; c < 3; ++c)
        rgba[c] = interpolate(e0[c], e1[c], cIdx, cBits);
      rgba[3] = interpolate(e0[3], e1[3], aIdx, aBits);
    }
    if(rotation)  // channel `rotation - 1` and alpha trade places
    {
      const int t       = rgba[3];
      rgba[3]           = rgba[rotation - 1];
      rgba[rotation - 1] = t;
    }
    for(int c = 0; c < 4; ++c)
      out[i][c] = uint8_t(rgba[c]);
  }
}

}  // namespace mihost
