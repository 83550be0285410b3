// ed_params.h -- kernel argument blocks (plain structs passed by value) and host-side launchers.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "edhip.h"

namespace ed {

constexpr int kMaxAxes = EDHIP_MAX_AXES;   // deformed axes on the GPU
constexpr int kMaxSteps = EDHIP_MAX_DIMS;  // non-deformed ("step") axes

// geometry shared by every input of one edhip_deform call: deform.c:381-391,439-451,771-776
struct GridGeom {
    int naxis;
    int has_affine;
    int disp_dtype;
    int pad_;
    int64_t in_len[kMaxAxes];     // I_k: deformed extents of inputs[0]          (deform.c:383)
    int64_t out_len[kMaxAxes];    // O_k: deformed extents of outputs[0]         (deform.c:384)
    int64_t off[kMaxAxes];        // crop offsets                                (deform.c:439-446)
    int64_t ncp[kMaxAxes];        // control points per axis                     (deform.c:449-451)
    const char* disp;             // prefiltered displacement grid
    int64_t disp_stride[kMaxAxes + 1];   // byte strides, [0] = component axis
    double affine[kMaxAxes * (kMaxAxes + 1)];   // inverse map, row-major naxis x (naxis+1)
    int64_t nvox;                 // prod O_k
};

// one input/output pair: deform.c:399-436,762-838
struct IOView {
    const char* in;               // forward: source; gradient: dX (accumulated into)
    char* out;                    // forward: destination; gradient: dY (read)
    int in_dtype, out_dtype;
    int order, mode;
    double cval;
    int64_t in_stride[kMaxAxes];  // byte strides of the deformed axes
    int64_t out_stride[kMaxAxes];
    int nstep;                    // number of non-deformed axes (ascending axis order)
    int steps_fastest;            // 1: flatten work as voxel*nsteps+step (step axes innermost in memory)
    int64_t nsteps;               // product of step extents
    int64_t step_len[kMaxSteps];
    int64_t in_step_stride[kMaxSteps];
    int64_t out_step_stride[kMaxSteps];
    // 16-bit float storage on the OUTPUT side of a float32 call (forward: the result is narrowed at the store;
    // gradient: dY is widened at the load): 1 half, 2 bfloat16.  out_dtype then reads EDHIP_F32 and every output
    // stride is twice its byte value, so that "stride / sizeof(float)" counts 16-bit elements.  Only the level-1
    // kernels of deform_hot.hip know about it (launch_tile declines every other route).
    int out16;
};

// launchers (defined in the .hip files, called from edhip_api.cpp); all enqueue on `stream` and
// return the hipError_t of the launch.
hipError_t launch_deform_exact(const GridGeom& g, const IOView& v, int gradient, hipStream_t stream);
// forward, 3 deformed axes, only the output voxels of a device-side list ([0] = count, [1 .. cap] =
// linear voxel ids; count > cap: every voxel): the near-tie voxels of the integer fast path
hipError_t launch_deform_exact_list(const GridGeom& g, const IOView& v, const int* list, int cap,
                                    hipStream_t stream);

// edhip_source_box: box[2h] = floor(min), box[2h+1] = ceil(max) of the raw (unmapped) source
// coordinate along axis h over every output voxel; `box` = 2 * naxis device ints
// conservative: the convex hull of the control coefficients (a superset of the exact box, O(grid points))
// sw (conservative only): also leave the FILTER window of one input on the device -- 2 * ndim ints (w0, w1) in
// the input's dimension order: the box widened by the tap window, the boundary mode's clipping, the filter's
// decay margin; last-axis rows on `align`-element boundaries; at least `minlen` samples along a deformed axis
struct SourceWindow {
    int32_t* out;                 // device, 2 * ndim ints (nullptr: none)
    int ndim;
    int shape[EDHIP_MAX_DIMS];
    int axis[kMaxAxes];           // array dimension of deformed axis h
    int order, mode, margin, align, minlen;
};
hipError_t launch_source_box(const GridGeom& g, int* box, hipStream_t stream, bool conservative = false,
                             const SourceWindow* sw = nullptr);

// fast path: returns hipErrorNotSupported (without launching) when the case is outside its
// envelope so that the caller can route it to the exact kernels instead.
hipError_t launch_deform_fast(const GridGeom& g, const IOView& v, int gradient, hipStream_t stream);
bool deform_fast_supported(const GridGeom& g, const IOView& v, int gradient);

// LDS-tiled forward kernel (3 deformed axes, order >= 2): the benchmark's hot kernel.
// A batch of independent volumes handled by ONE set of launches (edhip_deform_batch): sample b's
// arrays sit b * stride bytes after sample 0's; everything else (shapes, strides, order, mode, cval,
// crop, affine) is shared.  The strip / tile index of the kernels carries the sample.
struct GridPrefilter;
struct DeformBatch {
    int nbatch;
    int64_t in_bstride, out_bstride, disp_bstride;   // bytes
    int box_mode = 0;               // 1: EDHIP_FLAG_KEEP_BOXES (forward), 2: EDHIP_FLAG_USE_BOXES (gradient)
    const void* disp_id = nullptr;  // the caller's displacement pointer (identity of the control grid)
    int raw = 0;                    // EDHIP_FLAG_RAW_DISPLACEMENT was set
    int strong = 0;                 // EDHIP_FLAG_STRONG_FIELD was set
    // EDHIP_FLAG_ZERO_GRADIENT: the dense block to clear before the scatter; the tile path does it in its
    // tables launch and sets zero_done (left false: nothing has been launched that reads or adds to it)
    char* zero_ptr = nullptr;
    long long zero_bytes = 0;
    mutable bool zero_done = false;
    // EDHIP_FLAG_RAW_DISPLACEMENT, single volume: the control grid is still unfiltered -- the tables kernel of the tile
    // path filters it itself (every workgroup on its own LDS copy, workgroup 0 writes the filtered grid where
    // GridGeom::disp points) and sets gridpf_done; left false: nothing was launched, the caller filters it
    const GridPrefilter* gridpf = nullptr;
    mutable bool gridpf_done = false;
};
hipError_t launch_deform_tile(const GridGeom& g, const IOView& v, int gradient, hipStream_t stream,
                              const DeformBatch* batch = nullptr);
bool deform_tile_supported(const GridGeom& g, const IOView& v, int gradient);
size_t deform_tile_workspace_bytes(const GridGeom& g, int nbatch = 1, bool f64 = false);   // scratch the tile path will ask for (f64: a float64 volume is among the inputs)

// order-0 resampling of label maps (any dtype, 3 deformed axes, forward): bit-equal to the exact
// kernel (fast coordinates, exact re-evaluation of near-tie voxels), see deform_tile.hip
hipError_t launch_deform_label(const GridGeom& g, const IOView& v, hipStream_t stream, const DeformBatch* batch = nullptr);
bool deform_label_supported(const GridGeom& g, const IOView& v, int gradient);
// 8- / 16-bit integer volumes with spline orders 1-5 (forward): bit-equal to the exact kernel (fast
// coordinates and fp64 taps; voxels near a rounding tie or a coordinate boundary re-evaluated exactly)
hipError_t launch_deform_int(const GridGeom& g, const IOView& v, hipStream_t stream, const DeformBatch* batch = nullptr);
bool deform_int_supported(const GridGeom& g, const IOView& v, int gradient);
void tile_profile_enable(int enable);                    // edhip_profile_dominant
double tile_profile_last_us();                           // edhip_profile_last_us

struct FilterParams {
    const char* in;
    char* out;
    int in_dtype, out_dtype;
    int npoles;
    int transpose;
    double pole[2];
    double pole_pow[2];      // pole^(len-1), computed on the host with libm pow() like the reference
    int trunc_branch[2];     // transpose only: (int)ceil(log(1e-15)/log|p|) < len   (deform.c:1119,1134)
    double gain;
    int64_t len;             // extent along the filtered axis
    int64_t in_axis_stride, out_axis_stride;   // bytes
    int nouter;              // number of other axes
    int64_t nlines;
    int64_t outer_len[EDHIP_MAX_DIMS];
    int64_t in_outer_stride[EDHIP_MAX_DIMS];
    int64_t out_outer_stride[EDHIP_MAX_DIMS];
    double* ws;              // fp64 scratch for ws_lines lines, line-interleaved: ws[i * nl + j]
    int64_t ws_lines;        // lines per chunk
};

hipError_t launch_spline_filter(const FilterParams& p, hipStream_t stream);

// one-launch order-3 prefilter of a small control grid along every axis but the first
struct GridPrefilter {
    const char* in;
    char* out;               // contiguous, same dtype
    int dtype, elem_size;
    int ndim, total;
    int shape[kMaxAxes + 1];
    int64_t stride_bytes[kMaxAxes + 1];
    double pole, gain;
    double pole_pow[kMaxAxes + 1];   // pole^(shape[ax] - 1), host libm
    // EDHIP_FLAG_ZERO_GRADIENT: a dense, 16-byte-aligned block that workgroups 1 .. of the launch clear while
    // workgroup 0 filters the grid (the grid's recursion is one workgroup's latency: the rest of the chip is idle)
    char* zero_ptr;
    long long zero_bytes;
};
hipError_t launch_grid_prefilter(const GridPrefilter& p, hipStream_t stream);

// first bytes of every per-stream workspace are reserved for the prefiltered control grid
constexpr size_t kWorkspaceGridBytes = 64 * 1024;

// fast path (orders 2/3, float32/float64, lines >= 64 samples, no scratch); hipErrorNotSupported
// (nothing launched) when the case is outside its envelope
// window: 2 ints (w0, w1) per array dimension in DEVICE memory (edhip_source_window) -- only the samples inside
// are read and written; hipErrorNotSupported (nothing launched) when the whole-line tile kernels cannot take it
hipError_t launch_spline_filter_fast(const FilterParams& p, int order, int ndim, int axis,
                                     const int64_t* shape, const int64_t* in_stride_bytes,
                                     const int64_t* out_stride_bytes, hipStream_t stream,
                                     const int* window = nullptr, bool dry = false);

}  // namespace ed
