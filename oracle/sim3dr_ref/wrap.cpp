// C-linkage wrappers around the reference's own rasterizer, compiled IN PLACE from
// /root/reference/simple_romp/vis_human/sim3drender/lib/rasterize_kernel.cpp (see ../Makefile):
// the reference declares these functions with C++ linkage (rasterize.h:86-113) and calls them
// through Cython (rasterize.pyx); ctypes needs unmangled names.  Test infrastructure only.
#include "rasterize.h"

extern "C" {
void ref_get_normal(float* ver_normal, float* vertices, int* triangles, int nver, int ntri) {
    _get_normal(ver_normal, vertices, triangles, nver, ntri);
}
void ref_rasterize(unsigned char* image, float* vertices, int* triangles, float* colors, float* depth_buffer,
                   int ntri, int h, int w, int c, float alpha, int reverse) {
    _rasterize(image, vertices, triangles, colors, depth_buffer, ntri, h, w, c, alpha, reverse != 0);
}
}
