// Host-side frame input: PNG decode to cv2.imread's BGR u8 convention and a threaded read-ahead loader.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace bp {

struct IoError : std::runtime_error {
    explicit IoError(const std::string& m) : std::runtime_error(m) {}
};

// header only: size and the channel count of the stored image (palette counts as 3)
void png_info(const uint8_t* data, size_t n, int* h, int* w, int* channels);
// full decode into out[h*w*3] (B,G,R); scratch is reused between calls
void png_decode_bgr(const uint8_t* data, size_t n, uint8_t* out, size_t cap, int* h, int* w, std::vector<uint8_t>& scratch);
std::vector<uint8_t> read_file(const std::string& path);

class FrameLoader {
public:
    using HostAlloc = void* (*)(size_t);
    using HostFree = void (*)(void*);
    FrameLoader(std::vector<std::string> paths, int H, int W, int threads, int depth, HostAlloc alloc, HostFree free_fn);
    ~FrameLoader();
    FrameLoader(const FrameLoader&) = delete;
    FrameLoader& operator=(const FrameLoader&) = delete;
    // next frame in list order: 0 = ok (*bgr valid until release(*index)), 1 = list exhausted, -1 = this frame failed
    // (*err says why; the frame still has to be released)
    int next(long long* index, const uint8_t** bgr, std::string* err);
    void release(long long index);
    size_t size() const { return paths_.size(); }
    int height() const { return H_; }
    int width() const { return W_; }

private:
    struct Slot;
    void work();
    std::vector<std::string> paths_;
    int H_, W_;
    HostFree free_;
    std::vector<Slot> slots_;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_ready_, cv_free_;
    std::atomic<long long> next_job_{0};
    long long next_out_ = 0;
    bool stop_ = false;
};

// csrc/jpeg_bmp.cpp: baseline JPEG / uncompressed BMP -> interleaved RGB u8 (throws std::runtime_error)
void jpeg_decode_rgb(const uint8_t* data, size_t n, std::vector<uint8_t>& rgb, int* h, int* w);
void bmp_decode_rgb(const uint8_t* data, size_t n, std::vector<uint8_t>& rgb, int* h, int* w);

}  // namespace bp
