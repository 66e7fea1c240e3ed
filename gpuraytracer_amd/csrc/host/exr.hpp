// exr.hpp - minimal OpenEXR reader/writer (the reference uses tinyexr's LoadEXR / SaveEXR,
// Scene.cpp:81,110,504, which is not part of its repository).
//   write: float32 channels, 3 (B,G,R) or 4 (A,B,G,R), no compression (0) or ZIP (3, blocks of 16 lines);
//          exrCompressionFor() is the choice tinyexr's SaveEXR makes: ZIP unless both sides are under 16 pixels
//   read : single-part scanline and tiled files (the full-resolution level), NONE / RLE / ZIPS / ZIP / PIZ compression,
//          HALF or FLOAT channels named R,G,B,(A);
//          result is RGBA float32 (A = 1 when absent), row 0 = first scanline of the file
#pragma once

#include <string>
#include <vector>

namespace host {

bool writeEXR(const std::string &path, int width, int height, int channels, const float *pixels, std::string *err,
              int compression = 0);
inline int exrCompressionFor(int width, int height) { return (width < 16 && height < 16) ? 0 : 3; }
bool readEXR(const std::string &path, int &width, int &height, std::vector<float> &rgba, std::string *err);

}  // namespace host
