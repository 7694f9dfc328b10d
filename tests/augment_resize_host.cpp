// Test helper (never shipped): csrc/augment_resize.h -- the index / weight arithmetic the augmentor kernels execute -- compiled for the
// host, so that tests/test_host_cpu.py can compare it with oracle/augment_ref.py bit for bit without a GPU.
#define ODTK_HD inline
#include "augment_resize.h"

extern "C" {

float odtk_test_resize_scale(int n_in, int n_out) { return odtk::resize_scale_align(n_in, n_out); }

void odtk_test_nearest_indices(int n_in, int n_out, int* idx) {
    const float s = odtk::resize_scale_align(n_in, n_out);
    for (int o = 0; o < n_out; ++o) idx[o] = odtk::nearest_src(o, s, n_in);
}

void odtk_test_bicubic_taps(int n_in, int n_out, float* w, int* idx) {
    const float s = odtk::resize_scale_align(n_in, n_out);
    for (int o = 0; o < n_out; ++o) odtk::bicubic_taps((float)o * s, n_in, w + 4 * o, idx + 4 * o);
}

}
