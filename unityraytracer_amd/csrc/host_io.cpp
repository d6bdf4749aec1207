// host_io.cpp — host-side image I/O around the hot path.  Pure C++, no GPU.
//
//   urt_host_load_hdr     Radiance RGBE (.hdr) -> RGBA32F in the library's texture convention (row 0 = bottom): what
//                         `SkyboxTexture` needs (RS:9-10, RM:776).  The reference's sky assets (Assets/Skyboxes/*.hdr) are
//                         not in the reference tree (.MISSING_LARGE_BLOBS), so this is tested on synthetic files only.
//   urt_host_write_pfm    RGBA32F image -> .pfm (float RGB, bottom row first: the same row order as Result)
//   urt_host_write_png    RGBA32F linear image -> 8-bit sRGB .png, the kind of file RM:762's ScreenCapture.CaptureScreenshot
//                         writes ("Screenshots/<Time.time>-<_currentSample>.png"); stored (uncompressed) deflate blocks
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/urt.h"

namespace {

std::string g_io_error;
int io_fail(int code, const std::string& m) { g_io_error = m; return code; }

bool read_line(FILE* f, std::string& line) {
  line.clear();
  int c;
  while ((c = std::fgetc(f)) != EOF) {
    if (c == '\n') return true;
    line.push_back((char)c);
    if (line.size() > 4096) return false;
  }
  return !line.empty();
}

inline void rgbe_to_float(const unsigned char* p, float* out) {     // Ward's rgbe2float: mantissa * 2^(e - 136)
  if (p[3] == 0) { out[0] = out[1] = out[2] = 0.0f; return; }
  float f = std::ldexp(1.0f, (int)p[3] - (128 + 8));
  out[0] = (float)p[0] * f; out[1] = (float)p[1] * f; out[2] = (float)p[2] * f;
}

uint32_t crc_table[256];
bool crc_ready = false;
uint32_t crc32(uint32_t crc, const unsigned char* p, size_t n) {
  if (!crc_ready) {
    for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; crc_table[i] = c; }
    crc_ready = true;
  }
  crc = ~crc;
  for (size_t i = 0; i < n; i++) crc = crc_table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
  return ~crc;
}
void put32(std::vector<unsigned char>& v, uint32_t x) { v.push_back(x >> 24); v.push_back((x >> 16) & 0xff); v.push_back((x >> 8) & 0xff); v.push_back(x & 0xff); }
void chunk(std::vector<unsigned char>& png, const char* tag, const std::vector<unsigned char>& data) {
  put32(png, (uint32_t)data.size());
  size_t at = png.size();
  png.insert(png.end(), tag, tag + 4);
  png.insert(png.end(), data.begin(), data.end());
  put32(png, crc32(0, png.data() + at, 4 + data.size()));
}
inline unsigned char srgb8(float x) {           // linear -> sRGB transfer, clamped (NaN -> 0)
  if (!(x > 0.0f)) return 0;
  if (x >= 1.0f) return 255;
  float s = x <= 0.0031308f ? 12.92f * x : 1.055f * std::pow(x, 1.0f / 2.4f) - 0.055f;
  return (unsigned char)std::lround(s * 255.0f);
}

inline unsigned char unorm8(float a) {          // alpha: linear UNORM8
  if (!(a > 0.0f)) return 0;
  if (a >= 1.0f) return 255;
  return (unsigned char)std::lround(a * 255.0f);
}

}  // namespace

extern "C" {

// RGBA32F linear -> RGBA8: colour through the sRGB transfer function (the PNG writer's codes), alpha as UNORM8.  The GPU encoder of
// urt_texture_read_begin_format (csrc/present.hip) produces these bytes.
int urt_host_encode_srgb8(const float* rgba, size_t n_pixels, unsigned char* out_rgba8) {
  if ((!rgba || !out_rgba8) && n_pixels) return -1;
  for (size_t i = 0; i < n_pixels; i++) {
    out_rgba8[4 * i] = srgb8(rgba[4 * i]); out_rgba8[4 * i + 1] = srgb8(rgba[4 * i + 1]); out_rgba8[4 * i + 2] = srgb8(rgba[4 * i + 2]);
    out_rgba8[4 * i + 3] = unorm8(rgba[4 * i + 3]);
  }
  return 0;
}

// out[k], k = 1..255: the smallest float whose sRGB code is >= k (out[0] = -inf): the encoder as the step function it is.  Bisection
// over the bit patterns of the positive floats (ordered like the floats); the encoder is monotone — tests/test_host_io.py checks the
// table against the encoder around every step.
int urt_host_srgb8_first_floats(float* out256) {
  if (!out256) return -1;
  out256[0] = -std::numeric_limits<float>::infinity();
  for (int k = 1; k < 256; k++) {
    uint32_t lo = 0, hi = 0x3f800000u;               // code(+0) = 0 < k <= 255 = code(1.0f)
    while (hi - lo > 1) {
      uint32_t mid = lo + (hi - lo) / 2;
      float x; std::memcpy(&x, &mid, 4);
      if (srgb8(x) >= k) hi = mid; else lo = mid;
    }
    std::memcpy(&out256[k], &hi, 4);
  }
  return 0;
}

// Downscale to the importer's `maxTextureSize` (Assets/Skyboxes/*.hdr.meta:36 = 2048) with the resize filter the .meta names
// (platformSettings.resizeAlgorithm: 0 = Mitchell): separable Mitchell-Netravali (B = C = 1/3), the kernel widened by the scale
// factor (area filtering), edges clamped, weights normalised.  Unity's own resampler is closed source — its edge rule and the
// BC6H compression the .meta also asks for (textureCompression: 1) are NOT reproduced: an approximation of the import, unpinned.
static inline float mitchell(float x) {
  x = std::fabs(x);
  const float B = 1.0f / 3.0f, C = 1.0f / 3.0f;
  if (x < 1.0f) return ((12 - 9 * B - 6 * C) * x * x * x + (-18 + 12 * B + 6 * C) * x * x + (6 - 2 * B)) / 6.0f;
  if (x < 2.0f) return ((-B - 6 * C) * x * x * x + (6 * B + 30 * C) * x * x + (-12 * B - 48 * C) * x + (8 * B + 24 * C)) / 6.0f;
  return 0.0f;
}

static void resize_axis(const float* src, int n_src, int other, int n_dst, float* dst, bool along_x) {
  // along_x: src is [other rows][n_src cols][4], dst [other][n_dst][4]; else src is [n_src rows][other cols][4], dst [n_dst][other][4]
  float scale = (float)n_src / (float)n_dst, support = 2.0f * std::max(1.0f, scale), inv = 1.0f / std::max(1.0f, scale);
  std::vector<float> w;
  for (int d = 0; d < n_dst; d++) {
    float centre = ((float)d + 0.5f) * scale - 0.5f;
    int lo = (int)std::floor(centre - support), hi = (int)std::ceil(centre + support);
    w.assign((size_t)(hi - lo + 1), 0.0f);
    float sum = 0;
    for (int k = lo; k <= hi; k++) { float v = mitchell(((float)k - centre) * inv); w[(size_t)(k - lo)] = v; sum += v; }
    for (int o = 0; o < other; o++) {
      float acc[4] = {0, 0, 0, 0};
      for (int k = lo; k <= hi; k++) {
        int kk = std::min(std::max(k, 0), n_src - 1);
        const float* p = along_x ? src + 4 * ((size_t)o * n_src + kk) : src + 4 * ((size_t)kk * other + o);
        float wk = w[(size_t)(k - lo)] / sum;
        for (int c = 0; c < 4; c++) acc[c] += wk * p[c];
      }
      float* q = along_x ? dst + 4 * ((size_t)o * n_dst + d) : dst + 4 * ((size_t)d * other + o);
      for (int c = 0; c < 4; c++) q[c] = acc[c];
    }
  }
}

int urt_host_resize_rgba(const float* src, int width, int height, float* dst, int new_width, int new_height) {
  if (!src || !dst || width <= 0 || height <= 0 || new_width <= 0 || new_height <= 0) return io_fail(URT_ERR_INVALID_ARGUMENT, "resize: bad arguments");
  try {
    std::vector<float> tmp((size_t)new_width * height * 4);
    resize_axis(src, width, height, new_width, tmp.data(), true);
    resize_axis(tmp.data(), height, new_width, new_height, dst, false);
    return URT_OK;
  } catch (...) { return io_fail(URT_ERR_OUT_OF_MEMORY, "resize: allocation failed"); }
}

const char* urt_host_io_last_error(void) { return g_io_error.c_str(); }

int urt_host_load_hdr(const char* path, int* out_width, int* out_height, float* out_rgba, size_t capacity_floats) {
  if (!path || !out_width || !out_height) return io_fail(URT_ERR_INVALID_ARGUMENT, "load_hdr: NULL argument");
  FILE* f = std::fopen(path, "rb");
  if (!f) return io_fail(URT_ERR_INVALID_ARGUMENT, std::string("load_hdr: cannot open ") + path);
  std::string line;
  bool magic = false, fmt = false;
  int w = 0, h = 0;
  bool flip_y = true;   // "-Y H +X W": scanlines top to bottom
  while (read_line(f, line)) {
    if (!magic) { if (line.rfind("#?", 0) != 0) break; magic = true; continue; }
    if (line.rfind("FORMAT=32-bit_rle_rgbe", 0) == 0) fmt = true;
    if (line.empty()) {
      if (!read_line(f, line)) break;
      char sy = 0, sx = 0;
      if (std::sscanf(line.c_str(), "%cY %d %cX %d", &sy, &h, &sx, &w) != 4 || sx != '+' || (sy != '-' && sy != '+')) { w = h = 0; }
      flip_y = sy == '-';
      break;
    }
  }
  if (!magic || !fmt || w <= 0 || h <= 0 || w > 65536 || h > 65536) { std::fclose(f); return io_fail(URT_ERR_LAYOUT, "load_hdr: not a 32-bit_rle_rgbe Radiance file with a -Y H +X W resolution line"); }
  *out_width = w; *out_height = h;
  if (!out_rgba) { std::fclose(f); return URT_OK; }                         // size query
  if (capacity_floats < (size_t)w * h * 4) { std::fclose(f); return io_fail(URT_ERR_INVALID_ARGUMENT, "load_hdr: output buffer too small"); }
  std::vector<unsigned char> scan((size_t)w * 4);
  for (int y = 0; y < h; y++) {
    unsigned char hd[4];
    if (std::fread(hd, 1, 4, f) != 4) { std::fclose(f); return io_fail(URT_ERR_LAYOUT, "load_hdr: truncated file"); }
    if (w >= 8 && w < 32768 && hd[0] == 2 && hd[1] == 2 && ((hd[2] << 8) | hd[3]) == w) {
      for (int ch = 0; ch < 4; ch++) {                                     // new-style RLE: each channel separately
        int x = 0;
        while (x < w) {
          int c = std::fgetc(f);
          if (c == EOF) { std::fclose(f); return io_fail(URT_ERR_LAYOUT, "load_hdr: truncated scanline"); }
          if (c > 128) {                                                    // run
            int n = c - 128, v = std::fgetc(f);
            if (v == EOF || x + n > w) { std::fclose(f); return io_fail(URT_ERR_LAYOUT, "load_hdr: bad run"); }
            for (int k = 0; k < n; k++) scan[(size_t)(x++) * 4 + ch] = (unsigned char)v;
          } else {                                                          // literal
            int n = c;
            if (n == 0 || x + n > w) { std::fclose(f); return io_fail(URT_ERR_LAYOUT, "load_hdr: bad literal"); }
            for (int k = 0; k < n; k++) { int v = std::fgetc(f); if (v == EOF) { std::fclose(f); return io_fail(URT_ERR_LAYOUT, "load_hdr: truncated scanline"); } scan[(size_t)(x++) * 4 + ch] = (unsigned char)v; }
          }
        }
      }
    } else {                                                                // flat scanline (old-style RLE is not supported)
      std::memcpy(scan.data(), hd, 4);
      if (w > 1 && std::fread(scan.data() + 4, 1, (size_t)(w - 1) * 4, f) != (size_t)(w - 1) * 4) { std::fclose(f); return io_fail(URT_ERR_LAYOUT, "load_hdr: truncated flat scanline"); }
    }
    int row = flip_y ? h - 1 - y : y;                                       // library textures: row 0 = bottom
    float* dst = out_rgba + (size_t)row * w * 4;
    for (int x = 0; x < w; x++) { rgbe_to_float(scan.data() + (size_t)x * 4, dst + (size_t)x * 4); dst[(size_t)x * 4 + 3] = 1.0f; }
  }
  std::fclose(f);
  return URT_OK;
}

int urt_host_write_pfm(const char* path, const float* rgba, int width, int height) {
  if (!path || !rgba || width <= 0 || height <= 0) return io_fail(URT_ERR_INVALID_ARGUMENT, "write_pfm: bad arguments");
  FILE* f = std::fopen(path, "wb");
  if (!f) return io_fail(URT_ERR_INVALID_ARGUMENT, std::string("write_pfm: cannot open ") + path);
  std::fprintf(f, "PF\n%d %d\n-1.0\n", width, height);                      // negative scale = little endian; rows bottom to top
  std::vector<float> row((size_t)width * 3);
  for (int y = 0; y < height; y++) {
    const float* src = rgba + (size_t)y * width * 4;
    for (int x = 0; x < width; x++) { row[(size_t)x * 3] = src[(size_t)x * 4]; row[(size_t)x * 3 + 1] = src[(size_t)x * 4 + 1]; row[(size_t)x * 3 + 2] = src[(size_t)x * 4 + 2]; }
    if (std::fwrite(row.data(), sizeof(float), row.size(), f) != row.size()) { std::fclose(f); return io_fail(URT_ERR_INVALID_ARGUMENT, "write_pfm: short write"); }
  }
  std::fclose(f);
  return URT_OK;
}

int urt_host_write_png(const char* path, const float* rgba, int width, int height) {
  if (!path || !rgba || width <= 0 || height <= 0) return io_fail(URT_ERR_INVALID_ARGUMENT, "write_png: bad arguments");
  try {
    // raw image: filter byte 0 + RGB8 per row, TOP row first (PNG order) = our last row first
    std::vector<unsigned char> raw((size_t)height * (1 + (size_t)width * 3));
    size_t at = 0;
    for (int y = height - 1; y >= 0; y--) {
      raw[at++] = 0;
      const float* src = rgba + (size_t)y * width * 4;
      for (int x = 0; x < width; x++) { raw[at++] = srgb8(src[(size_t)x * 4]); raw[at++] = srgb8(src[(size_t)x * 4 + 1]); raw[at++] = srgb8(src[(size_t)x * 4 + 2]); }
    }
    std::vector<unsigned char> z;                                           // zlib stream of stored blocks
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t pos = 0; pos < raw.size();) {
      size_t n = std::min<size_t>(65535, raw.size() - pos);
      z.push_back(pos + n == raw.size() ? 1 : 0);
      z.push_back(n & 0xff); z.push_back((n >> 8) & 0xff); z.push_back(~n & 0xff); z.push_back((~n >> 8) & 0xff);
      z.insert(z.end(), raw.begin() + (long)pos, raw.begin() + (long)(pos + n));
      for (size_t i = 0; i < n; i++) { a = (a + raw[pos + i]) % 65521u; b = (b + a) % 65521u; }
      pos += n;
    }
    put32(z, (b << 16) | a);
    std::vector<unsigned char> png = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<unsigned char> ihdr;
    put32(ihdr, (uint32_t)width); put32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8-bit RGB
    chunk(png, "IHDR", ihdr);
    chunk(png, "IDAT", z);
    chunk(png, "IEND", {});
    FILE* f = std::fopen(path, "wb");
    if (!f) return io_fail(URT_ERR_INVALID_ARGUMENT, std::string("write_png: cannot open ") + path);
    bool ok = std::fwrite(png.data(), 1, png.size(), f) == png.size();
    std::fclose(f);
    return ok ? URT_OK : io_fail(URT_ERR_INVALID_ARGUMENT, "write_png: short write");
  } catch (...) { return io_fail(URT_ERR_OUT_OF_MEMORY, "write_png: allocation failed"); }
}

}  // extern "C"
