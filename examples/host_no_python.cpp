// host_no_python.cpp -- the C ABI used the way a host WITHOUT Python uses it: a plan file (python -m romp_amd.export), a batch
// of pre-processed frames, romp_net_load -> romp_net_forward -> romp_parse.  tests/test_gpu_parity.py builds and runs it and
// compares its output files with the Python path, byte for byte.
//   hipcc --offload-arch=gfx950 -O2 -I include examples/host_no_python.cpp -L romp_amd -lromp_hip -Wl,-rpath,$PWD/romp_amd -o host_no_python
//   ./host_no_python romp.plan frames.f32 B center_thresh out_prefix
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "romp_hip.h"

#define CHECK(call)                                                                      \
    do {                                                                                 \
        const int rc_ = (call);                                                          \
        if (rc_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, romp_last_error()); return 1; } \
    } while (0)

static bool write_file(const char* prefix, const char* suffix, const void* data, size_t bytes) {
    char path[1024];
    snprintf(path, sizeof(path), "%s%s", prefix, suffix);
    FILE* f = fopen(path, "wb");
    if (!f) return false;
    const bool ok = fwrite(data, 1, bytes, f) == bytes;
    fclose(f);
    return ok;
}

int main(int argc, char** argv) {
    if (argc != 6) { fprintf(stderr, "usage: %s plan frames.f32 B center_thresh out_prefix\n", argv[0]); return 2; }
    const int B = atoi(argv[3]);
    const float thresh = (float)atof(argv[4]);
    if (romp_abi_version() != ROMP_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    romp_net* net = nullptr;
    CHECK(romp_net_load(&net, argv[1], B));
    int32_t size = 0, n_ops = 0;
    int64_t cf = 0, pf = 0;
    CHECK(romp_net_plan_info(net, &size, &cf, &pf, &n_ops));
    int32_t split_k_items = 0;
    CHECK(romp_net_plan_kind(net, &split_k_items));
    if (split_k_items > 0 && B > 2) fprintf(stderr, "note: %s is a single-image plan (split_k_items %d); export a batch plan for B = %d\n", argv[1], split_k_items, B);
    const size_t img_floats = (size_t)B * size * size * 3;
    std::vector<float> frames(img_floats);
    FILE* f = fopen(argv[2], "rb");
    if (!f || fread(frames.data(), 4, img_floats, f) != img_floats) { fprintf(stderr, "cannot read %zu floats from %s\n", img_floats, argv[2]); return 1; }
    fclose(f);
    hipStream_t st;
    if (hipStreamCreate(&st) != hipSuccess) return 1;
    float *d_img, *d_center, *d_params;
    const int max_person = 64, cap = B * max_person;
    if (hipMalloc((void**)&d_img, img_floats * 4) != hipSuccess || hipMalloc((void**)&d_center, (size_t)B * cf * 4) != hipSuccess ||
        hipMalloc((void**)&d_params, (size_t)B * pf * 4) != hipSuccess) return 1;
    hipMemcpy(d_img, frames.data(), img_floats * 4, hipMemcpyHostToDevice);
    CHECK(romp_net_set_graph(net, 1));
    for (int rep = 0; rep < 2; ++rep)                       // second call replays the captured graph
        CHECK(romp_net_forward(net, d_img, B, d_center, d_params, st));
    int32_t *d_i32;                                         // batch_ids, flat_inds, center_preds (2 per person), workspace
    float* d_f32;                                           // scores, params_pred, cam, thetas, betas
    const size_t n_i32 = (size_t)cap * 4 + (size_t)B * (2 * max_person + 2), n_f32 = (size_t)cap * (1 + 145 + 3 + 72 + 10);
    if (hipMalloc((void**)&d_i32, n_i32 * 4) != hipSuccess || hipMalloc((void**)&d_f32, n_f32 * 4) != hipSuccess) return 1;
    int32_t count = 0;
    float *scores = d_f32, *params_pred = scores + cap, *cam = params_pred + (size_t)cap * 145, *thetas = cam + (size_t)cap * 3,
          *betas = thetas + (size_t)cap * 72;
    CHECK(romp_parse(d_center, d_params, B, thresh, max_person, &count, d_i32, d_i32 + cap, scores, params_pred, cam, thetas, betas,
                     d_i32 + 2 * cap, d_i32 + 4 * cap, st));
    hipStreamSynchronize(st);
    std::vector<float> center((size_t)B * cf), th((size_t)count * 72), cm((size_t)count * 3);
    std::vector<int32_t> flat(count);
    hipMemcpy(center.data(), d_center, center.size() * 4, hipMemcpyDeviceToHost);
    if (count) {
        hipMemcpy(th.data(), thetas, th.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(cm.data(), cam, cm.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(flat.data(), d_i32 + cap, flat.size() * 4, hipMemcpyDeviceToHost);
    }
    if (!write_file(argv[5], ".center.f32", center.data(), center.size() * 4) || !write_file(argv[5], ".thetas.f32", th.data(), th.size() * 4) ||
        !write_file(argv[5], ".cam.f32", cm.data(), cm.size() * 4) || !write_file(argv[5], ".flat.i32", flat.data(), flat.size() * 4)) return 1;
    printf("plan %s: %d ops, input %dx%d, batch %d -> %d persons\n", argv[1], n_ops, size, size, B, count);
    romp_net_destroy(net);
    return 0;
}
