// ed_workspace.h -- per-(device, stream) cached scratch buffers.
//
// The kernels need a little device scratch per call (spill worklists, the displacement tables of
// the tile kernels, the fp64 line buffer of the exact prefilter).  hipMallocAsync / hipFreeAsync
// per call costs hundreds of microseconds of host time on ROCm 7.2 -- more than the kernels -- so
// scratch is kept: one buffer per (device, stream), grown on demand, reused by every later call on
// that stream.  Work on one stream is ordered, so consecutive calls can share the buffer; growing
// synchronises that stream once before the old buffer is released.  Buffers live until the
// process exits or workspace_release_all() (edhip_release_scratch) is called.  The library stays re-entrant: the table is mutex-protected and calls on
// different streams never share a buffer.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>

namespace ed {

// Returns a device pointer to at least `bytes` bytes of scratch owned by (current device, stream),
// or nullptr with *err set.  The contents are unspecified.
void* workspace_reserve(hipStream_t stream, size_t bytes, hipError_t* err);

// A second, small cached buffer per (device, stream) that nothing but its owner writes: the tile
// bounding boxes a forward deform leaves for the gradient call that follows it (deform_tile.hip).
// `key` is a host-side blob describing what the buffer holds (all zero = nothing); it is reset when
// the buffer moves.  Callers hold the stream's StreamGuard.
struct KeepKey {
    unsigned long long w[48];
};
void* keep_reserve(hipStream_t stream, size_t bytes, KeepKey** key, hipError_t* err);

// Level-1 spill feedback of the tile path, per (device, stream).  The tile kernels count the tiles whose
// source box does not fit the standard (small) level-1 LDS box; the tables kernel of the NEXT call on
// the stream reports that count into pinned host memory (one 8-byte store: sequence number << 32 |
// count) and nobody waits for it.  The host reads whatever has arrived when it prepares a call and
// picks the larger level-1 boxes (one workgroup per CU fewer) for geometries whose recent calls spilled
// more than a tenth of their tiles.  A hint about speed only: results do not depend on the box size.
// Callers hold the stream's StreamGuard.
struct SpillHint {
    volatile unsigned long long* host = nullptr;   // pinned + mapped
    unsigned long long* dev = nullptr;             // the device's address of *host
    unsigned seq = 0;                              // last sequence number handed out (never 0)
    struct Call { unsigned seq; unsigned long long key; unsigned tiles; } ring[8] = {};
    struct Entry { unsigned long long key; float frac; } table[8] = {};
    int nring = 0, ntable = 0;

    void absorb();                                  // fold the latest report into the table
    float fraction(unsigned long long key) const;   // last known spilled fraction for a geometry, or 0
    bool known(unsigned long long key) const;       // a report for this geometry has arrived
    unsigned begin_call(unsigned long long key, unsigned tiles);   // -> this call's sequence number
};
SpillHint* spill_hint(hipStream_t stream);          // nullptr when pinned memory is not available

// What the head of a stream's workspace holds: the filtered copy of which raw control grid
// (EDHIP_FLAG_RAW_DISPLACEMENT), written where.  A call with EDHIP_FLAG_GRID_STAYS whose raw grid matches
// the stamp -- and whose workspace has not moved -- skips the prefilter launch.  Callers hold the StreamGuard.
struct GridStamp {
    const void* raw = nullptr;
    const void* ws = nullptr;
    unsigned long long generation = 0;    // workspace_generation() when the copy was made
    int dtype = 0, ndim = 0;
    long long shape[9] = {}, stride[9] = {};
};
GridStamp* grid_stamp(hipStream_t stream);
// counts the times the stream's workspace buffer was freed or replaced (a stamp from an earlier generation is void)
unsigned long long workspace_generation(hipStream_t stream);

// A second stream per (device, stream) with the two events that fork it from / join it to the caller's stream, for the
// forward route of deform_k1z.hip: the general-tile kernel runs next to the class-A kernel.  `parity` alternates
// between the calls of that route (which pair of work-list counters in the workspace head the call uses).  nullptr:
// no second stream could be made (the caller launches everything on its own stream).  Callers hold the StreamGuard.
struct SideLane {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    unsigned parity = 0;
    bool usable = false;
    // the geometry stage's own buffer (tables, tile records, work lists + their counters): nothing else writes it
    void* geo_ptr = nullptr;
    size_t geo_cap = 0;
};
SideLane* side_lane(hipStream_t stream);
// at least `bytes` of the lane's geometry buffer (grown on demand: both streams are drained first);
// the first 4 KiB -- the work lists' counters -- are cleared when the buffer is allocated
void* geo_reserve(hipStream_t stream, SideLane* lane, size_t bytes, hipError_t* err);

// Drains the devices that own scratch and frees every cached buffer.
void workspace_release_all();

// Requests above 320 MiB are served by a stream-ordered allocation (hipMallocAsync) that is not
// cached: workspace_trim() -- called when an API call has enqueued its work -- releases it with
// hipFreeAsync behind the kernels that use it.
void workspace_trim(hipStream_t stream);

// Serialises the host side of API calls that target the same (device, stream): the workspace of a
// stream is shared by consecutive calls (control grid, tables, spill lists), so two host threads
// enqueueing on ONE stream -- torch's default stream is shared by all Python threads, and ctypes
// drops the GIL -- must not interleave their launches (thread A's tables, thread B's tables, A's
// tile kernel reading B's tables), nor may one grow the buffer while the other still holds the old
// pointer.  Calls only enqueue work, so the lock is held for microseconds; calls on different
// streams or devices never contend.  Re-entrant (edhip_deform_batch calls edhip_deform).
class StreamGuard {
public:
    explicit StreamGuard(hipStream_t stream);
    ~StreamGuard();
    StreamGuard(const StreamGuard&) = delete;
    StreamGuard& operator=(const StreamGuard&) = delete;

private:
    void* mutex_;
    hipStream_t stream_;
};

}  // namespace ed
