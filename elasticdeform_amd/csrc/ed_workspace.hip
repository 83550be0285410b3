#include "ed_workspace.h"

#include <map>
#include <memory>
#include <mutex>
#include <utility>

namespace ed {

namespace {
struct Buffer {
    void* ptr = nullptr;
    size_t cap = 0;
};
std::mutex g_mutex;
std::map<std::pair<int, hipStream_t>, Buffer> g_buffers;
// one host-side lock per (device, stream); entries are never erased, so the pointers stay valid
std::map<std::pair<int, hipStream_t>, std::unique_ptr<std::recursive_mutex>> g_stream_locks;
}  // namespace

StreamGuard::StreamGuard(hipStream_t stream) : mutex_(nullptr)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::recursive_mutex* m;
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        auto& slot = g_stream_locks[std::make_pair(dev, stream)];
        if (!slot)
            slot.reset(new std::recursive_mutex);
        m = slot.get();
    }
    m->lock();
    mutex_ = m;
}

StreamGuard::~StreamGuard()
{
    if (mutex_)
        static_cast<std::recursive_mutex*>(mutex_)->unlock();
}

void* workspace_reserve(hipStream_t stream, size_t bytes, hipError_t* err)
{
    *err = hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        *err = e;
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_mutex);
    Buffer& b = g_buffers[std::make_pair(dev, stream)];
    if (b.cap >= bytes && b.ptr)
        return b.ptr;
    if (b.ptr) {
        // earlier kernels on this stream may still be using the old buffer
        e = hipStreamSynchronize(stream);
        if (e == hipSuccess)
            e = hipFree(b.ptr);
        b.ptr = nullptr;
        b.cap = 0;
        if (e != hipSuccess) {
            *err = e;
            return nullptr;
        }
    }
    // grow geometrically, 1 MiB granularity
    size_t want = bytes + bytes / 4;
    want = (want + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    e = hipMalloc(&b.ptr, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipMalloc(&b.ptr, bytes);
        want = bytes;
    }
    if (e != hipSuccess) {
        b.ptr = nullptr;
        *err = e;
        return nullptr;
    }
    b.cap = want;
    return b.ptr;
}

void workspace_release_all()
{
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    std::lock_guard<std::mutex> lock(g_mutex);
    for (auto& kv : g_buffers) {
        if (!kv.second.ptr)
            continue;
        // everything enqueued on that device must be done with the buffer (the stream itself may
        // already have been destroyed by its owner, so the whole device is drained)
        if (hipSetDevice(kv.first.first) == hipSuccess) {
            (void)hipDeviceSynchronize();
            (void)hipFree(kv.second.ptr);
        }
        kv.second.ptr = nullptr;
        kv.second.cap = 0;
    }
    g_buffers.clear();
    if (have_cur)
        (void)hipSetDevice(cur);
    (void)hipGetLastError();
}

}  // namespace ed
