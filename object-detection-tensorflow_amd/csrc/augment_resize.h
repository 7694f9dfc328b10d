// Source-index and weight arithmetic of the augmentor's NEAREST_NEIGHBOR / BICUBIC zoom (csrc/augment.hip), kept in a header that also
// compiles as plain C++: tests/test_host_cpu.py builds it with g++ (ODTK_HD empty) and checks every index and weight against
// oracle/augment_ref.py bit for bit without a GPU.  Reference: utils/image_augmentor.py:72-76, :98-101, :125-128 ->
// tf.image.resize_images(..., align_corners=True) -> ResizeNearestNeighbor / ResizeBicubic of TensorFlow 1.13.
#ifndef ODTK_AUGMENT_RESIZE_H_
#define ODTK_AUGMENT_RESIZE_H_
#include <math.h>
#ifndef ODTK_HD
#define ODTK_HD __host__ __device__ __forceinline__
#endif

namespace odtk {

// CalculateResizeScale (image_resizer_state.h) with align_corners: (in - 1) / float(out - 1) for out > 1; with out == 1 the only
// position is 0 and the scale never matters
ODTK_HD float resize_scale_align(int n_in, int n_out) { return n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f; }

// TF 1.13 ResizeNearestNeighbor with align_corners: in = min(roundf(out * scale), in_size - 1)
ODTK_HD int nearest_src(int out_pos, float scale, int n_in) {
    const int i = (int)roundf((float)out_pos * scale);
    return i < n_in - 1 ? i : n_in - 1;
}

// TF 1.13 ResizeBicubic (resize_bicubic_op.cc): Keys kernel, A = -0.75, read from a 1 025-entry table at lrintf(delta * 1024);
// an entry is the polynomial evaluated in double on a float abscissa and stored as float -- recomputed here instead of a table
// (k / 1024 and 1 + k / 1024 are exact floats).  Taps floor - 1 .. floor + 2 clamped to the picture; along x first
// (v0 w0 + v1 w1 + v2 w2 + v3 w3 in float, no contraction), then the four row results along y.
ODTK_HD void bicubic_taps(float pos, int limit, float* w, int* idx) {
    const int loc = (int)pos;
    const float delta = pos - (float)loc;
    const int off = (int)rintf(delta * 1024.f);
    const double A = -0.75;
    const double t = (double)((float)off * (1.f / 1024.f)), u = (double)((float)(1024 - off) * (1.f / 1024.f));
    const double t1 = t + 1.0, u1 = u + 1.0;
    w[0] = (float)(((A * t1 - 5.0 * A) * t1 + 8.0 * A) * t1 - 4.0 * A);
    w[1] = (float)(((A + 2.0) * t - (A + 3.0)) * t * t + 1.0);
    w[2] = (float)(((A + 2.0) * u - (A + 3.0)) * u * u + 1.0);
    w[3] = (float)(((A * u1 - 5.0 * A) * u1 + 8.0 * A) * u1 - 4.0 * A);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = loc - 1 + k;
        idx[k] = i < 0 ? 0 : (i > limit - 1 ? limit - 1 : i);
    }
}

}  // namespace odtk
#endif  // ODTK_AUGMENT_RESIZE_H_
