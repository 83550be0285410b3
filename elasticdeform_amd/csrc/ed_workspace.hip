#include "ed_workspace.h"

#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <utility>

namespace ed {

namespace {
struct Buffer {
    void* ptr = nullptr;
    size_t cap = 0;
    bool transient = false;      // allocated stream-ordered for one oversized call; freed by workspace_trim
    // bumped whenever `ptr` is freed or replaced: what a GridStamp remembers about the head of the buffer is void
    // then, even if the allocator hands the same address back (the new buffer's head is cleared)
    unsigned long long generation = 0;
};
// Requests above this are not kept: the float32 order-4/5 cascade asks for a dense temporary of the
// array's size (up to 1 GiB here) and the exact filter for 256 MiB; cached per stream they stay
// pinned until edhip_release_scratch and a job with many streams can run out of memory.
constexpr size_t kKeepBytes = (size_t)320 << 20;
std::mutex g_mutex;
std::map<std::pair<int, hipStream_t>, Buffer> g_buffers;
struct KeepBuffer {
    void* ptr = nullptr;
    size_t cap = 0;
    KeepKey key;
};
std::map<std::pair<int, hipStream_t>, KeepBuffer> g_keep;
struct HintSlot {
    SpillHint h;
    bool tried = false;
};
std::map<std::pair<int, hipStream_t>, HintSlot> g_hints;      // (std::map: addresses are stable)
struct SideSlot {
    SideLane lane;
    bool tried = false;
};
std::map<std::pair<int, hipStream_t>, SideSlot> g_sides;      // (std::map: addresses are stable; never erased)
// one host-side lock per (device, stream); entries are never erased, so the pointers stay valid
std::map<std::pair<int, hipStream_t>, std::unique_ptr<std::recursive_mutex>> g_stream_locks;
}  // namespace

StreamGuard::StreamGuard(hipStream_t stream) : mutex_(nullptr), stream_(stream)
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
    if (mutex_) {
        workspace_trim(stream_);         // an oversized scratch buffer does not outlive its call
        static_cast<std::recursive_mutex*>(mutex_)->unlock();
    }
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
        if (b.transient) {
            e = hipFreeAsync(b.ptr, stream);      // stream-ordered behind the kernels that used it
        } else {
            // earlier kernels on this stream may still be using the old buffer: the one place where
            // the library waits for the stream (documented in edhip.h)
            e = hipStreamSynchronize(stream);
            if (e == hipSuccess)
                e = hipFree(b.ptr);
        }
        b.ptr = nullptr;
        b.cap = 0;
        b.transient = false;
        ++b.generation;
        if (e != hipSuccess) {
            *err = e;
            return nullptr;
        }
    }
    ++b.generation;          // (a new allocation follows)
    if (bytes > kKeepBytes) {
        // oversized, one call only: stream-ordered allocation, released by workspace_trim() when the
        // call has enqueued its work (no host synchronisation, nothing stays pinned)
        e = hipMallocAsync(&b.ptr, bytes, stream);
        if (e == hipSuccess) {
            b.cap = bytes;
            b.transient = true;
            (void)hipMemsetAsync(b.ptr, 0, (size_t)65536, stream);      // (spill-feedback words: see below)
            return b.ptr;
        }
        (void)hipGetLastError();
        b.ptr = nullptr;             // fall through to the plain allocation
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
    // the head holds the spill-feedback words (count, sequence number) the next tables kernel reports to the
    // host: leftover memory must not pass for a report (SpillHint::absorb matches small sequence numbers)
    (void)hipMemsetAsync(b.ptr, 0, want < (size_t)65536 ? want : (size_t)65536, stream);
    return b.ptr;
}

SideLane* side_lane(hipStream_t stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        return nullptr;
    std::lock_guard<std::mutex> lock(g_mutex);
    SideSlot& sl = g_sides[std::make_pair(dev, stream)];
    if (!sl.tried) {
        sl.tried = true;
        hipStream_t s2 = nullptr;
        hipEvent_t e1 = nullptr, e2 = nullptr;
        // (default priority: at the lowest one the general-tile kernel was starved until the class-A kernel had finished,
        // 110 -> 167 us)
        if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&e1, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&e2, hipEventDisableTiming) == hipSuccess) {
            sl.lane.stream = s2;
            sl.lane.fork = e1;
            sl.lane.join = e2;
            sl.lane.usable = true;
        } else {
            (void)hipGetLastError();
            if (e1)
                (void)hipEventDestroy(e1);
            if (s2)
                (void)hipStreamDestroy(s2);
        }
    }
    return &sl.lane;
}

void* geo_reserve(hipStream_t stream, SideLane* lane, size_t bytes, hipError_t* err)
{
    *err = hipSuccess;
    std::lock_guard<std::mutex> lock(g_mutex);
    if (lane->geo_ptr && lane->geo_cap >= bytes)
        return lane->geo_ptr;
    hipError_t e = hipSuccess;
    if (lane->geo_ptr) {
        // kernels of earlier calls (either stream) may still be using the old buffer
        e = hipStreamSynchronize(stream);
        if (e == hipSuccess && lane->stream)
            e = hipStreamSynchronize(lane->stream);
        if (e == hipSuccess)
            e = hipFree(lane->geo_ptr);
        lane->geo_ptr = nullptr;
        lane->geo_cap = 0;
        if (e != hipSuccess) {
            *err = e;
            return nullptr;
        }
    }
    size_t want = bytes + bytes / 4;
    want = (want + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    e = hipMalloc(&lane->geo_ptr, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        lane->geo_ptr = nullptr;
        *err = e;
        return nullptr;
    }
    lane->geo_cap = want;
    e = hipMemsetAsync(lane->geo_ptr, 0, 4096, stream);
    if (e != hipSuccess) {
        *err = e;
        return nullptr;
    }
    return lane->geo_ptr;
}

void* keep_reserve(hipStream_t stream, size_t bytes, KeepKey** key, hipError_t* err)
{
    *err = hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        *err = e;
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_mutex);
    KeepBuffer& b = g_keep[std::make_pair(dev, stream)];      // (std::map: the address is stable)
    *key = &b.key;
    if (b.ptr && b.cap >= bytes)
        return b.ptr;
    if (b.ptr) {
        e = hipStreamSynchronize(stream);        // earlier kernels may still read the old buffer
        if (e == hipSuccess)
            e = hipFree(b.ptr);
        b.ptr = nullptr;
        b.cap = 0;
        if (e != hipSuccess) {
            *err = e;
            return nullptr;
        }
    }
    b.key = KeepKey();
    size_t want = (bytes + bytes / 4 + 65535) & ~(size_t)65535;
    e = hipMalloc(&b.ptr, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        b.ptr = nullptr;
        *err = e;
        return nullptr;
    }
    b.cap = want;
    return b.ptr;
}

GridStamp* grid_stamp(hipStream_t stream)
{
    static std::map<std::pair<int, hipStream_t>, GridStamp> stamps;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        dev = -1;
    std::lock_guard<std::mutex> lock(g_mutex);
    return &stamps[std::make_pair(dev, stream)];      // (std::map: the address is stable)
}

SpillHint* spill_hint(hipStream_t stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        return nullptr;
    std::lock_guard<std::mutex> lock(g_mutex);
    HintSlot& s = g_hints[std::make_pair(dev, stream)];
    if (!s.tried) {
        s.tried = true;
        void* hp = nullptr;
        void* dp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess && hp &&
            hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess && dp) {
            memset(hp, 0, 64);
            s.h.host = static_cast<volatile unsigned long long*>(hp);
            s.h.dev = static_cast<unsigned long long*>(dp);
        } else {
            if (hp)
                (void)hipHostFree(hp);
            (void)hipGetLastError();
        }
    }
    return s.h.host ? &s.h : nullptr;
}

void SpillHint::absorb()
{
    const unsigned long long v = *host;
    const unsigned s = (unsigned)(v >> 32), count = (unsigned)v;
    if (s == 0)
        return;
    for (int i = 0; i < nring; ++i) {
        if (ring[i].seq != s)
            continue;
        const float f = ring[i].tiles ? (float)count / (float)ring[i].tiles : 0.f;
        int slot = -1;
        for (int t = 0; t < ntable; ++t)
            if (table[t].key == ring[i].key)
                slot = t;
        if (slot < 0) {
            if (ntable < 8)
                slot = ntable++;
            else {                      // (eight geometries in rotation on one stream: forget the oldest)
                for (int t = 1; t < 8; ++t)
                    table[t - 1] = table[t];
                slot = 7;
            }
            table[slot].key = ring[i].key;
        }
        table[slot].frac = f;
        ring[i].seq = 0;                // consumed
        return;
    }
}

float SpillHint::fraction(unsigned long long key) const
{
    for (int t = 0; t < ntable; ++t)
        if (table[t].key == key)
            return table[t].frac;
    return 0.f;
}

bool SpillHint::known(unsigned long long key) const
{
    for (int t = 0; t < ntable; ++t)
        if (table[t].key == key)
            return true;
    return false;
}

unsigned SpillHint::begin_call(unsigned long long key, unsigned tiles)
{
    seq = seq + 1 ? seq + 1 : 1;
    if (nring < 8)
        ++nring;
    for (int i = nring - 1; i > 0; --i)
        ring[i] = ring[i - 1];
    ring[0].seq = seq;
    ring[0].key = key;
    ring[0].tiles = tiles;
    return seq;
}

void workspace_trim(hipStream_t stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        return;
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_buffers.find(std::make_pair(dev, stream));
    if (it == g_buffers.end() || !it->second.ptr || !it->second.transient)
        return;
    (void)hipFreeAsync(it->second.ptr, stream);
    it->second.ptr = nullptr;
    it->second.cap = 0;
    it->second.transient = false;
    ++it->second.generation;
}

unsigned long long workspace_generation(hipStream_t stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        return 0;
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_buffers.find(std::make_pair(dev, stream));
    return it == g_buffers.end() ? 0 : it->second.generation;
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
            if (kv.second.transient)
                (void)hipFreeAsync(kv.second.ptr, kv.first.second);
            else
                (void)hipFree(kv.second.ptr);
        }
        kv.second.ptr = nullptr;
        kv.second.cap = 0;
        kv.second.transient = false;
        ++kv.second.generation;          // (entries stay: a GridStamp taken before the release must not match again)
    }
    for (auto& kv : g_keep) {
        if (kv.second.ptr && hipSetDevice(kv.first.first) == hipSuccess) {
            (void)hipDeviceSynchronize();
            (void)hipFree(kv.second.ptr);
        }
        kv.second.ptr = nullptr;
        kv.second.cap = 0;
        kv.second.key = KeepKey();       // (entries stay: callers may hold the key's address)
    }
    for (auto& kv : g_sides) {
        if (kv.second.lane.geo_ptr && hipSetDevice(kv.first.first) == hipSuccess) {
            (void)hipDeviceSynchronize();
            (void)hipFree(kv.second.lane.geo_ptr);
        }
        kv.second.lane.geo_ptr = nullptr;
        kv.second.lane.geo_cap = 0;
        // (the second stream and its events stay)
    }
    for (auto& kv : g_hints) {
        if (kv.second.h.host && hipSetDevice(kv.first.first) == hipSuccess) {
            (void)hipDeviceSynchronize();
            (void)hipHostFree(const_cast<unsigned long long*>(kv.second.h.host));
        }
        kv.second = HintSlot();          // (entries stay; the slot is allocated again on demand)
    }
    if (have_cur)
        (void)hipSetDevice(cur);
    (void)hipGetLastError();
}

}  // namespace ed
