// main.cpp — the reference's command line (src/main.rs:236-258) as a native program over the C ABI:
//
//     rttnw <scene>          scene number 1..9 with the reference's per-scene size, spp and camera (main.rs:66-183),
//                            writes image.png into the current directory (main.rs:231) and prints the wall time.
//
// Host side only: it builds the scene with the scenes.rs mirror (scenes.cpp, through the library's entry-point table),
// renders with rttnw_render / rttnw_render_multi and encodes the RGBA8 framebuffer.  The PNG codec below (8-bit RGB /
// RGBA, non-interlaced — what assets/earth.png is and what main.rs:230-231 writes) sits on zlib, standing in for the
// `image` crate of the reference's host.  Extras that are not in the reference: --width --spp --out --precision --seed
// --gpus --assets.  No CPU fallback: without a HIP device the commit fails and the program exits like the reference's
// DummyError path.
#include "../../include/rttnw_hip.h"
#include "../../include/rttnw_scenes.h"

#include <zlib.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" const rttnw_builder_api* rttnw_builder(void); // the table scenes.cpp is written against (include/rttnw_hip.h)

namespace {

uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | uint32_t(p[3]); }
void put_be32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(uint8_t(x >> 24)); v.push_back(uint8_t(x >> 16)); v.push_back(uint8_t(x >> 8)); v.push_back(uint8_t(x)); }

// Decode an 8-bit RGB / RGBA non-interlaced PNG into RGBA8, top row first.  False on anything else (the caller then passes
// NULL to the library: the reference's cyan fallback for a texture that failed to load, texture.rs:102-105).
bool png_read_rgba8(const std::string& path, std::vector<uint8_t>& rgba, uint32_t& w, uint32_t& h) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<uint8_t> file;
    uint8_t buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + n);
    std::fclose(f);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (file.size() < 33 || std::memcmp(file.data(), sig, 8) != 0) return false;
    std::vector<uint8_t> idat;
    uint32_t channels = 0;
    for (size_t pos = 8; pos + 12 <= file.size();) {
        const uint32_t len = be32(&file[pos]);
        const uint8_t* type = &file[pos + 4];
        if (pos + 12 + size_t(len) > file.size()) return false;
        const uint8_t* data = &file[pos + 8];
        if (std::memcmp(type, "IHDR", 4) == 0) {
            if (len != 13) return false;
            w = be32(data); h = be32(data + 4);
            const uint8_t depth = data[8], colour = data[9], interlace = data[12];
            if (depth != 8 || interlace != 0 || (colour != 2 && colour != 6) || w == 0 || h == 0 || w > 65535 || h > 65535) return false;
            channels = colour == 2 ? 3 : 4;
        } else if (std::memcmp(type, "IDAT", 4) == 0) {
            idat.insert(idat.end(), data, data + len);
        } else if (std::memcmp(type, "IEND", 4) == 0) {
            break;
        }
        pos += 12 + size_t(len);
    }
    if (!channels || idat.empty()) return false;
    const size_t stride = size_t(w) * channels;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf raw_len = uLongf(raw.size());
    if (uncompress(raw.data(), &raw_len, idat.data(), uLong(idat.size())) != Z_OK || raw_len != raw.size()) return false;
    rgba.assign(size_t(w) * h * 4, 255);
    std::vector<uint8_t> prev(stride, 0), cur(stride);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t filter = raw[(stride + 1) * y];
        const uint8_t* in = &raw[(stride + 1) * y + 1];
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= channels ? cur[i - channels] : 0, b = prev[i], c = i >= channels ? prev[i - channels] : 0;
            int pred = 0;
            switch (filter) {
            case 0: pred = 0; break;
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            case 4: {
                const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                break;
            }
            default: return false;
            }
            cur[i] = uint8_t(in[i] + pred);
        }
        for (uint32_t x = 0; x < w; ++x)
            for (uint32_t k = 0; k < channels; ++k) rgba[(size_t(y) * w + x) * 4 + k] = cur[size_t(x) * channels + k];
        prev.swap(cur);
    }
    return true;
}

void png_chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data) {
    put_be32(out, uint32_t(data.size()));
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put_be32(out, uint32_t(crc32(0L, &out[start], uInt(out.size() - start))));
}
// image::save_buffer(path, buffer, width, height, ColorType::Rgba8) — main.rs:230-231
bool png_write_rgba8(const std::string& path, const uint8_t* rgba, uint32_t w, uint32_t h) {
    std::vector<uint8_t> raw;
    raw.reserve((size_t(w) * 4 + 1) * h);
    for (uint32_t y = 0; y < h; ++y) {
        raw.push_back(0); // filter: none
        raw.insert(raw.end(), rgba + size_t(y) * w * 4, rgba + size_t(y + 1) * w * 4);
    }
    uLongf zlen = compressBound(uLong(raw.size()));
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), uLong(raw.size()), 6) != Z_OK) return false;
    z.resize(zlen);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'}, ihdr;
    put_be32(ihdr, w); put_be32(ihdr, h);
    const uint8_t tail[5] = {8, 6, 0, 0, 0}; // 8 bits, RGBA, deflate, adaptive filtering, no interlace
    ihdr.insert(ihdr.end(), tail, tail + 5);
    png_chunk(out, "IHDR", ihdr);
    png_chunk(out, "IDAT", z);
    png_chunk(out, "IEND", {});
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    return std::fclose(f) == 0 && ok;
}

int usage(const char* argv0) { // main.rs:238-250
    std::fprintf(stderr, "Usage: %s <scene>\nPossible scenes:\n", argv0);
    for (uint32_t n = 1; n <= 9; ++n) std::fprintf(stderr, "\t- %u: %s\n", n, rttnw_scenes_name(n));
    std::fprintf(stderr, "There was an error\n"); // DummyError, main.rs:260-268
    return 1;
}

} // namespace

int main(int argc, char** argv) {
    long scene_number = -1;
    uint32_t width = 0, spp = 0, gpus = 1;
    uint64_t seed = 1;
    uint32_t precision = RTTNW_F64; // --precision f64 (default: the reference's arithmetic) | f32 (throughput) | f64strict (nothing contracted: the CPU reference's path decisions bit for bit)
    std::string out = "image.png", assets;
    { // next to the executable: rttnw_amd/host/ -> rttnw_amd/assets/
        const std::string self(argv[0]);
        const size_t slash = self.find_last_of('/');
        assets = (slash == std::string::npos ? std::string(".") : self.substr(0, slash)) + "/../assets";
    }
    if (argc == 4 && std::strcmp(argv[1], "--reencode") == 0) { // codec check (tests): decode argv[2], encode it as argv[3]
        std::vector<uint8_t> px;
        uint32_t w = 0, h = 0;
        return png_read_rgba8(argv[2], px, w, h) && png_write_rgba8(argv[3], px.data(), w, h) ? 0 : 1;
    }
    for (int i = 1; i < argc; ++i) {
        const std::string a(argv[i]);
        auto value = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--width") width = uint32_t(std::atoi(value()));
        else if (a == "--spp") spp = uint32_t(std::atoi(value()));
        else if (a == "--gpus") gpus = uint32_t(std::atoi(value()));
        else if (a == "--seed") seed = std::strtoull(value(), nullptr, 10);
        else if (a == "--out") out = value();
        else if (a == "--assets") assets = value();
        else if (a == "--precision") {
            const char* v = value();
            precision = std::strcmp(v, "f32") == 0 ? RTTNW_F32 : (std::strcmp(v, "f64strict") == 0 ? RTTNW_F64_STRICT : RTTNW_F64);
        }
        else if (!a.empty() && a[0] != '-' && scene_number < 0) {
            char* end = nullptr;
            scene_number = std::strtol(a.c_str(), &end, 10);
            if (*end || scene_number < 0) return usage(argv[0]); // `.parse().map_err(|_| ERROR)?` — main.rs:252
        } else return usage(argv[0]);
    }
    if (scene_number < 0 || gpus == 0 || gpus > 64) return usage(argv[0]);

    const char* name = rttnw_scenes_name(uint32_t(scene_number));
    std::printf("Scene number: %ld\n", scene_number); // main.rs:253
    if (!name) {
        std::fprintf(stderr, "There is no scene %ld\n", scene_number); // main.rs:179-182
        return 1;
    }
    std::printf("Running scene %s\n", name);
    const auto t0 = std::chrono::steady_clock::now();

    std::vector<uint8_t> earth;
    uint32_t ew = 0, eh = 0;
    const bool have_earth = png_read_rgba8(assets + "/earth.png", earth, ew, eh);
    if (!have_earth && (scene_number == 4 || scene_number == 9))
        std::fprintf(stderr, "note: %s/earth.png not readable: the image texture is cyan, like texture.rs:102-105\n", assets.c_str());

    const uint64_t scene_seed = 0x5eed0001ull;
    rttnw_scene* scene = nullptr;
    rttnw_scene_setup setup;
    int rc = rttnw_scene_create(scene_seed, &scene);
    if (rc == RTTNW_OK)
        rc = rttnw_scenes_build(rttnw_builder(), scene, name, scene_seed, have_earth ? earth.data() : nullptr, ew, eh, 0, &setup);
    if (rc != RTTNW_OK) {
        std::fprintf(stderr, "%s\nThere was an error\n", rttnw_last_error());
        return 1;
    }

    rttnw_params p;
    std::memset(&p, 0, sizeof p);
    const double aspect = double(setup.width) / double(setup.height);
    p.width = width ? width : setup.width;
    p.height = uint32_t(double(p.width) / aspect); // `(width as f64 / aspect_ratio) as u32` — main.rs:184
    p.spp = spp ? spp : setup.spp;
    p.max_depth = 50;   // main.rs:216
    p.t_min = 0.001;    // main.rs:33
    for (int k = 0; k < 3; ++k) p.background[k] = setup.background[k];
    p.seed = seed;
    p.precision = precision;
    p.quirks = RTTNW_QUIRKS_REFERENCE;
    p.tile_world = 1;

    std::vector<uint8_t> rgba(size_t(p.width) * p.height * 4);
    std::vector<rttnw_stats> stats(gpus);
    if (gpus == 1) {
        rc = rttnw_render(scene, &setup.camera, &p, nullptr, rgba.data(), stats.data());
    } else {
        const int nd = rttnw_device_count();
        std::vector<int32_t> dev(gpus);
        for (uint32_t r = 0; r < gpus; ++r) dev[r] = int32_t(r % uint32_t(nd > 0 ? nd : 1)); // fewer devices than ranks: logical ranks
        rc = rttnw_render_multi(scene, &setup.camera, &p, gpus, dev.data(), nullptr, rgba.data(), stats.data());
    }
    if (rc != RTTNW_OK) {
        std::fprintf(stderr, "%s\nThere was an error\n", rttnw_last_error());
        return 1;
    }
    if (!png_write_rgba8(out, rgba.data(), p.width, p.height)) {
        std::fprintf(stderr, "cannot write %s\nThere was an error\n", out.c_str());
        return 1;
    }
    rttnw_scene_destroy(scene);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double device_ms = 0.0;
    for (const auto& s : stats) device_ms = device_ms > s.kernel_ms ? device_ms : s.kernel_ms;
    std::printf("%.3fs (%ux%u, %u samples per pixel, %s kernels on %u GPU(s): %.1f ms device time, %.1f Msamples/s)\n", secs, p.width, p.height,
                p.spp, precision == RTTNW_F32 ? "f32" : (precision == RTTNW_F64_STRICT ? "f64strict" : "f64"), gpus, device_ms, double(p.width) * p.height * p.spp / (device_ms > 0 ? device_ms : 1e-9) / 1e3);
    return 0;
}
