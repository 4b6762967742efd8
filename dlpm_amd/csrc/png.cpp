// png.cpp -- host-side PNG writer for the generated-image dump (8-bit RGB, non-interlaced).
//
// Replaces the per-sample torchvision.utils.save_image -> PIL call of bem/evaluate/EvaluationManager.py:188-190
// with a native encoder (zlib deflate, per-row adaptive filter chosen by the minimum-sum-of-absolute-differences
// heuristic) and a small thread pool, so that a chunk's files are written while the next chunk is sampling.
// The decoded pixels are what matters for the FID/PRDC readers downstream; the byte stream is a valid PNG but is
// not meant to equal PIL's.
#include <zlib.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "common.h"

namespace dlpm {
namespace {

inline void put32(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)(v >> 24);
    p[1] = (uint8_t)(v >> 16);
    p[2] = (uint8_t)(v >> 8);
    p[3] = (uint8_t)v;
}

// writes one chunk at dst, returns its size
inline int64_t chunk(uint8_t *dst, const char *type, const uint8_t *data, uint32_t len) {
    put32(dst, len);
    memcpy(dst + 4, type, 4);
    if (len) memcpy(dst + 8, data, len);
    uint32_t c = (uint32_t)crc32(0L, dst + 4, len + 4);
    put32(dst + 8 + len, c);
    return (int64_t)len + 12;
}

inline int paeth(int a, int b, int c) {
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// filter one row with `type` into out[0..rowbytes); returns the heuristic cost
int64_t filter_row(int type, const uint8_t *cur, const uint8_t *prev, int rowbytes, uint8_t *out) {
    const int bpp = 3;
    int64_t cost = 0;
    for (int i = 0; i < rowbytes; ++i) {
        int a = i >= bpp ? cur[i - bpp] : 0;
        int b = prev ? prev[i] : 0;
        int c = (prev && i >= bpp) ? prev[i - bpp] : 0;
        int pred = 0;
        switch (type) {
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            case 4: pred = paeth(a, b, c); break;
            default: break;
        }
        uint8_t v = (uint8_t)(cur[i] - pred);
        out[i] = v;
        cost += v < 128 ? v : 256 - v;
    }
    return cost;
}

int encode(const uint8_t *hwc, int H, int W, int level, uint8_t *out, int64_t cap, int64_t *len) {
    const int rowbytes = 3 * W;
    std::vector<uint8_t> raw((size_t)H * (rowbytes + 1));
    std::vector<uint8_t> cand((size_t)rowbytes);
    for (int y = 0; y < H; ++y) {
        const uint8_t *cur = hwc + (size_t)y * rowbytes;
        const uint8_t *prev = y ? cur - rowbytes : nullptr;
        uint8_t *dst = raw.data() + (size_t)y * (rowbytes + 1);
        int64_t best = -1;
        for (int t = 0; t < 5; ++t) {
            int64_t c = filter_row(t, cur, prev, rowbytes, cand.data());
            if (best < 0 || c < best) {
                best = c;
                dst[0] = (uint8_t)t;
                memcpy(dst + 1, cand.data(), rowbytes);
            }
        }
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z((size_t)zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), level) != Z_OK) {
        set_error("dlpm_png_encode_rgb8: zlib compress2 failed");
        return DLPM_ERR_ARG;
    }
    int64_t need = 8 + (13 + 12) + ((int64_t)zlen + 12) + 12;
    if (need > cap) {
        set_error("dlpm_png_encode_rgb8: output buffer too small (%lld > %lld)", (long long)need, (long long)cap);
        return DLPM_ERR_ARG;
    }
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    memcpy(out, sig, 8);
    int64_t o = 8;
    uint8_t ihdr[13];
    put32(ihdr, (uint32_t)W);
    put32(ihdr + 4, (uint32_t)H);
    ihdr[8] = 8;   // bit depth
    ihdr[9] = 2;   // colour type: truecolour
    ihdr[10] = 0;  // deflate
    ihdr[11] = 0;  // adaptive filtering
    ihdr[12] = 0;  // no interlace
    o += chunk(out + o, "IHDR", ihdr, 13);
    o += chunk(out + o, "IDAT", z.data(), (uint32_t)zlen);
    o += chunk(out + o, "IEND", nullptr, 0);
    *len = o;
    return DLPM_OK;
}

}  // namespace
}  // namespace dlpm

using namespace dlpm;

extern "C" int64_t dlpm_png_bound(int32_t H, int32_t W) {
    if (H <= 0 || W <= 0) return -1;
    return 8 + 25 + 12 + 12 + (int64_t)compressBound((uLong)H * (3 * (uLong)W + 1));
}

extern "C" int dlpm_png_encode_rgb8(const uint8_t *hwc, int32_t H, int32_t W, int32_t level, uint8_t *out, int64_t cap,
                                    int64_t *len) {
    DLPM_CHECK_ARG(hwc && out && len && H > 0 && W > 0 && level >= 0 && level <= 9, "dlpm_png_encode_rgb8: bad argument");
    return encode(hwc, H, W, level, out, cap, len);
}

extern "C" int dlpm_png_write_rgb8(const uint8_t *hwc_batch, int64_t B, int32_t H, int32_t W, const char *dir,
                                   int64_t first_index, int32_t level, int32_t nthreads) {
    DLPM_CHECK_ARG(hwc_batch && dir && B > 0 && H > 0 && W > 0 && level >= 0 && level <= 9 && first_index >= 0,
                   "dlpm_png_write_rgb8: bad argument");
    int nt = nthreads > 0 ? nthreads : 1;
    if (nt > B) nt = (int)B;
    std::atomic<int64_t> next(0);
    std::atomic<int> failed(0);
    std::string first_err;
    const int64_t cap = dlpm_png_bound(H, W);
    const size_t img = (size_t)H * W * 3;
    auto work = [&]() {
        std::vector<uint8_t> buf((size_t)cap);
        for (;;) {
            int64_t i = next.fetch_add(1);
            if (i >= B || failed.load()) break;
            int64_t len = 0;
            if (encode(hwc_batch + (size_t)i * img, H, W, level, buf.data(), cap, &len) != DLPM_OK) {
                failed.store(1);
                break;
            }
            std::string path = std::string(dir) + "/" + std::to_string(first_index + i) + ".png";   // f"{i}.png"
            FILE *f = fopen(path.c_str(), "wb");
            if (!f || fwrite(buf.data(), 1, (size_t)len, f) != (size_t)len) {
                if (f) fclose(f);
                if (!failed.exchange(1)) first_err = "dlpm_png_write_rgb8: cannot write " + path;
                break;
            }
            fclose(f);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    if (failed.load()) {
        set_error("%s", first_err.empty() ? "dlpm_png_write_rgb8: PNG encoding failed" : first_err.c_str());
        return first_err.empty() ? DLPM_ERR_ARG : DLPM_ERR_IO;
    }
    return DLPM_OK;
}
