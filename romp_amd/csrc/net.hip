// net.hip -- executor of the layer program (seam #1 of include/romp_hip.h).
//
// The host lowers the model definition (HRNet-32 + ROMP head, model.py:246-481) to a flat list of
// romp_op; this file owns the activation arena (NHWC float32, sized for max_batch, resident in HBM
// for the life of the context) and replays the list on the caller's stream -- eagerly, or from a
// hipGraph captured per (batch, I/O pointers) so that the ~350 dependent launches of one forward
// cost one graph launch on the host.
#include "common.h"
#include <vector>
#include <map>
#include <tuple>
#include <string.h>

namespace romp {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace romp

using namespace romp;

struct GraphKey {
    int B; const void* img; void* center; void* params;
    bool operator<(const GraphKey& o) const {
        return std::tie(B, img, center, params) < std::tie(o.B, o.img, o.center, o.params);
    }
};

struct romp_net {
    std::vector<romp_op> ops;
    std::vector<int64_t> buf_floats;     // per image
    std::vector<float*> bufs;
    int max_batch = 0;
    int mode = 0;
    int use_graph = 0;
    std::map<GraphKey, hipGraphExec_t> graphs;
};

static const float* resolve_in(romp_net* n, int buf, const float* image) {
    if (buf == ROMP_BUF_IMAGE) return image;
    if (buf >= 0 && buf < (int)n->bufs.size()) return n->bufs[buf];
    return nullptr;
}
static float* resolve_out(romp_net* n, int buf, float* center, float* params) {
    if (buf == ROMP_BUF_CENTER) return center;
    if (buf == ROMP_BUF_PARAMS) return params;
    if (buf >= 0 && buf < (int)n->bufs.size()) return n->bufs[buf];
    return nullptr;
}

static int run_op(romp_net* n, const romp_op& op, const float* image, int B, float* center, float* params,
                  hipStream_t st) {
    switch (op.kind) {
        case ROMP_OP_STEM: {
            const float* in = resolve_in(n, op.in_buf, image);
            float* out = resolve_out(n, op.out_buf, center, params);
            ROMP_REQUIRE(in && out, "stem: bad buffers %d -> %d", op.in_buf, op.out_buf);
            return launch_stem(op, in, out, B, st);
        }
        case ROMP_OP_CONV: {
            const float* in = resolve_in(n, op.in_buf, image);
            float* out = resolve_out(n, op.out_buf, center, params);
            const float* res = op.res_buf == ROMP_BUF_NONE ? nullptr : resolve_in(n, op.res_buf, image);
            ROMP_REQUIRE(in && out, "conv: bad buffers %d -> %d", op.in_buf, op.out_buf);
            ROMP_REQUIRE(op.res_buf == ROMP_BUF_NONE || res, "conv: bad residual buffer %d", op.res_buf);
            return launch_conv(op, in, res, out, B, n->mode, st);
        }
        case ROMP_OP_FUSESUM: {
            FuseTerm t[4];
            ROMP_REQUIRE(op.n_terms >= 1 && op.n_terms <= 4, "fusesum: n_terms %d", op.n_terms);
            for (int k = 0; k < op.n_terms; ++k) {
                t[k].ptr = resolve_in(n, op.term_buf[k], image);
                ROMP_REQUIRE(t[k].ptr, "fusesum: bad term buffer %d", op.term_buf[k]);
                t[k].shift = op.term_shift[k];
                t[k].cstride = op.term_cstride[k];
            }
            float* out = resolve_out(n, op.out_buf, center, params);
            ROMP_REQUIRE(out, "fusesum: bad out buffer %d", op.out_buf);
            return launch_fusesum(t, op.n_terms, out, B, op.H, op.W, op.Cout, op.out_cstride, op.out_coff, op.relu, st);
        }
        default:
            set_error("unknown op kind %d", op.kind);
            return ROMP_EINVAL;
    }
}

static int run_all(romp_net* n, const float* image, int B, float* center, float* params, hipStream_t st) {
    for (size_t i = 0; i < n->ops.size(); ++i) {
        int rc = run_op(n, n->ops[i], image, B, center, params, st);
        if (rc) return rc;
    }
    return ROMP_OK;
}

extern "C" {

int romp_abi_version(void) { return ROMP_ABI_VERSION; }
const char* romp_last_error(void) { return romp::g_err; }

int romp_net_create(romp_net** out, const romp_op* ops_host, int n_ops, const int64_t* buf_floats, int n_bufs,
                    int max_batch) {
    ROMP_REQUIRE(out && ops_host && n_ops > 0 && n_bufs >= 0 && max_batch > 0, "romp_net_create: bad arguments");
    romp_net* n = new romp_net();
    n->ops.assign(ops_host, ops_host + n_ops);
    n->buf_floats.assign(buf_floats, buf_floats + n_bufs);
    n->max_batch = max_batch;
    n->bufs.resize(n_bufs, nullptr);
    for (int i = 0; i < n_bufs; ++i) {
        const size_t bytes = (size_t)buf_floats[i] * max_batch * sizeof(float);
        hipError_t e = hipMalloc((void**)&n->bufs[i], bytes);
        if (e != hipSuccess) {
            set_error("arena buffer %d: hipMalloc(%zu) failed: %s", i, bytes, hipGetErrorString(e));
            romp_net_destroy(n);
            return ROMP_ENOMEM;
        }
        // padded channels (e.g. the head input's coord/zero channels) must start defined
        e = hipMemset(n->bufs[i], 0, bytes);
        if (e != hipSuccess) { set_error("hipMemset failed: %s", hipGetErrorString(e)); romp_net_destroy(n); return ROMP_EHIP; }
    }
    *out = n;
    return ROMP_OK;
}

int romp_net_set_mode(romp_net* n, int mode) {
    ROMP_REQUIRE(n && (mode == 0 || mode == 1), "romp_net_set_mode: bad arguments");
    n->mode = mode;
    return ROMP_OK;
}

int romp_net_set_graph(romp_net* n, int enable) {
    ROMP_REQUIRE(n, "romp_net_set_graph: null net");
    n->use_graph = enable ? 1 : 0;
    return ROMP_OK;
}

int romp_net_forward(romp_net* n, const float* image, int B, float* center, float* params, void* stream) {
    ROMP_REQUIRE(n && image && center && params && B > 0, "romp_net_forward: bad arguments");
    if (B > n->max_batch) { set_error("batch %d > max_batch %d", B, n->max_batch); return ROMP_ECAPACITY; }
    hipStream_t st = (hipStream_t)stream;
    if (!n->use_graph || n->mode != 0) return run_all(n, image, B, center, params, st);
    GraphKey key{B, image, center, params};
    auto it = n->graphs.find(key);
    if (it == n->graphs.end()) {
        ROMP_REQUIRE(st != nullptr, "graph mode needs a non-default stream");
        hipGraph_t g = nullptr;
        ROMP_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        int rc = run_all(n, image, B, center, params, st);
        hipError_t e = hipStreamEndCapture(st, &g);
        if (rc) { if (g) hipGraphDestroy(g); return rc; }
        ROMP_HIP_CHECK(e);
        hipGraphExec_t ge = nullptr;
        ROMP_HIP_CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipGraphDestroy(g);
        it = n->graphs.emplace(key, ge).first;
    }
    ROMP_HIP_CHECK(hipGraphLaunch(it->second, st));
    return ROMP_OK;
}

int romp_net_profile(romp_net* n, const float* image, int B, float* center, float* params, void* stream,
                     float* ms_out, int iters) {
    ROMP_REQUIRE(n && image && center && params && ms_out && B > 0 && iters > 0, "romp_net_profile: bad arguments");
    if (B > n->max_batch) { set_error("batch %d > max_batch %d", B, n->max_batch); return ROMP_ECAPACITY; }
    hipStream_t st = (hipStream_t)stream;
    const size_t nops = n->ops.size();
    std::vector<hipEvent_t> ev(nops + 1);
    for (auto& e : ev) ROMP_HIP_CHECK(hipEventCreate(&e));
    for (size_t i = 0; i < nops; ++i) ms_out[i] = 0.f;
    int rc = ROMP_OK;
    for (int it = 0; it < iters && rc == ROMP_OK; ++it) {
        hipEventRecord(ev[0], st);
        for (size_t i = 0; i < nops; ++i) {
            rc = run_op(n, n->ops[i], image, B, center, params, st);
            if (rc) break;
            hipEventRecord(ev[i + 1], st);
        }
        if (rc) break;
        if (hipStreamSynchronize(st) != hipSuccess) { set_error("hipStreamSynchronize failed"); rc = ROMP_EHIP; break; }
        for (size_t i = 0; i < nops; ++i) {
            float ms = 0.f;
            hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            ms_out[i] += ms / iters;
        }
    }
    for (auto& e : ev) hipEventDestroy(e);
    return rc;
}

int romp_net_read_buffer(romp_net* n, int buf, int B, float* dst, int64_t n_floats, void* stream) {
    ROMP_REQUIRE(n && dst && buf >= 0 && buf < (int)n->bufs.size(), "romp_net_read_buffer: bad buffer %d", buf);
    ROMP_REQUIRE(B > 0 && B <= n->max_batch && n_floats <= n->buf_floats[buf] * B, "romp_net_read_buffer: bad size");
    ROMP_HIP_CHECK(hipMemcpyAsync(dst, n->bufs[buf], (size_t)n_floats * sizeof(float), hipMemcpyDeviceToDevice,
                                  (hipStream_t)stream));
    return ROMP_OK;
}

int romp_net_write_buffer(romp_net* n, int buf, const float* src, int64_t n_floats, void* stream) {
    ROMP_REQUIRE(n && src && buf >= 0 && buf < (int)n->bufs.size(), "romp_net_write_buffer: bad buffer %d", buf);
    ROMP_REQUIRE(n_floats <= n->buf_floats[buf] * n->max_batch, "romp_net_write_buffer: bad size");
    ROMP_HIP_CHECK(hipMemcpyAsync(n->bufs[buf], src, (size_t)n_floats * sizeof(float), hipMemcpyDeviceToDevice,
                                  (hipStream_t)stream));
    return ROMP_OK;
}

void romp_net_destroy(romp_net* n) {
    if (!n) return;
    for (auto& kv : n->graphs) hipGraphExecDestroy(kv.second);
    for (float* p : n->bufs)
        if (p) hipFree(p);
    delete n;
}

int romp_conv_forward(const romp_op* op, const float* in, const float* res, float* out, int B, int mode, void* stream) {
    ROMP_REQUIRE(op && in && out && B > 0, "romp_conv_forward: bad arguments");
    if (op->kind == ROMP_OP_STEM) return launch_stem(*op, in, out, B, (hipStream_t)stream);
    ROMP_REQUIRE(op->kind == ROMP_OP_CONV, "romp_conv_forward: op kind %d", op->kind);
    return launch_conv(*op, in, res, out, B, mode, (hipStream_t)stream);
}

int romp_conv_describe(const romp_op* op, int B, char* out, int n) {
    ROMP_REQUIRE(op && out && n > 0 && B > 0, "romp_conv_describe: bad arguments");
    if (op->kind == ROMP_OP_STEM) { snprintf(out, n, "stem_conv"); return ROMP_OK; }
    if (op->kind == ROMP_OP_FUSESUM) { snprintf(out, n, "fusesum"); return ROMP_OK; }
    return describe_conv(*op, B, out, n);
}

}  // extern "C"
