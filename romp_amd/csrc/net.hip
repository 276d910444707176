// net.hip -- executor of the layer program (seam #1 of include/romp_hip.h).
//
// The host lowers the model definition (HRNet-32 + ROMP head, model.py:246-481) to a flat list of
// romp_op; this file owns the activation arena (NHWC float32, sized for max_batch, resident in HBM
// for the life of the context), the per-op work queues of the persistent conv kernels, the
// per-batch-size kernel-variant table filled by romp_net_autotune, and replays the list on the
// caller's stream and up to three side streams (FORK / JOIN regions, RECORD / WAIT edges inside them) -- eagerly, or from a hipGraph
// per (batch, output pointers; the stem runs eagerly in front of it) that build_graph assembles node by node with the program's
// dependencies, so that the ~220-330 dependent launches of one forward cost one graph launch on the host.
#include "common.h"
#include "conv_common.h"   // h2_unpack (range scan)
#include <algorithm>
#include <vector>
#include <map>
#include <tuple>
#include <string.h>
#include <stdio.h>
#include <stdint.h>
#include <unistd.h>

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
    int B; const void* img; void* center; void* params; int lane;      // lane -1: whole forward in one graph
    bool operator<(const GraphKey& o) const {
        return std::tie(B, img, center, params, lane) < std::tie(o.B, o.img, o.center, o.params, o.lane);
    }
};

struct romp_net {
    std::vector<romp_op> ops;
    std::vector<int64_t> buf_floats;     // per image
    std::vector<float*> bufs;
    int* queues = nullptr;               // 2 lanes x n_ops x QUEUE_INTS work counters, zeroed at the start of a forward
    int* sat = nullptr;                  // saturation counter (conv_common.h sat_report): cumulative since create / the last reset
    bool sat_checked = false;            // also launch the counting builds of the fused-block kernels (env ROMP_CHECK_FINITE=1, range scan)
    int max_batch = 0;
    int mode = 0;
    int use_graph = 0;
    bool image_only_in_op0 = false;      // ops[0] is a stem and no later op reads ROMP_BUF_IMAGE (checked at create): graph replay may then
                                         // launch the stem eagerly and key the graph without the image pointer
    int eager_ops = 1;                   // ops of that eager prefix: 1 (a stem), 2 (the NOP holding the stem + ROMP_OP_STEM2)
    int use_streams = 1;                 // run FORK/JOIN regions on side streams
    // Batch lanes: with split == 2 a forward of B images runs as two independent half-batch op sequences
    // on two streams (lane 0 on the caller's stream), each conv capped at one workgroup per CU, so the
    // two lanes' kernels co-reside on every CU in different phases: one lane's synchronized epilogue /
    // prologue memory bursts run under the other lane's MFMA phase instead of stalling the chip.
    int split = 1;
    int wg_cap = 0;                      // workgroups per CU a conv may take (0: all it can)
    int64_t image_floats = 0, center_floats = 0, params_floats = 0;   // per image, for the lane offsets
    hipStream_t lane_main = nullptr;     // lane 1's main stream
    hipStream_t scratch = nullptr;       // build_graph: the stream every op is captured on, alone
    hipStream_t side[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    hipEvent_t ev_fork[2] = {nullptr, nullptr};
    hipEvent_t ev_join[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    hipEvent_t ev_begin = nullptr, ev_offset = nullptr, ev_done = nullptr;
    std::vector<hipEvent_t> ev_edge[2];  // ROMP_OP_RECORD / ROMP_OP_WAIT events, per lane
    std::map<int, std::vector<int>> tuned;   // batch -> variant per op (-1: heuristic)
    std::map<GraphKey, hipGraphExec_t> graphs;
    // a net loaded from a plan file (romp_net_load) owns its constants; one built by romp_net_create borrows the host's
    void* plan_dev = nullptr;
    std::vector<char> plan_host;
    int32_t plan_input_size = 0;
    int64_t plan_center_floats = 0, plan_params_floats = 0;
    int32_t plan_split_k_items = 0;      // plan kind recorded in the file: > 0 = single-image plan (export.py)
};

// Arena buffers are batch-major: image b of buffer `buf` starts at b * buf_floats[buf].  `b0` is the
// first image of the lane being run (0 unless the forward is split into batch lanes); the caller's
// image / center / params pointers are already offset.
static const float* resolve_in(romp_net* n, int buf, const float* image, int b0) {
    if (buf == ROMP_BUF_IMAGE) return image;
    if (buf >= 0 && buf < (int)n->bufs.size()) return n->bufs[buf] + (size_t)b0 * n->buf_floats[buf];
    return nullptr;
}
static float* resolve_out(romp_net* n, int buf, float* center, float* params, int b0) {
    if (buf == ROMP_BUF_CENTER) return center;
    if (buf == ROMP_BUF_PARAMS) return params;
    if (buf >= 0 && buf < (int)n->bufs.size()) return n->bufs[buf] + (size_t)b0 * n->buf_floats[buf];
    return nullptr;
}

static int run_op(romp_net* n, size_t idx, int variant, const float* image, int B, float* center, float* params,
                  hipStream_t st, int lane = 0, int b0 = 0) {
    const romp_op& op = n->ops[idx];
    int* queue = n->queues + ((size_t)lane * n->ops.size() + idx) * QUEUE_INTS;
    auto resolve_in = [&](romp_net* nn, int buf, const float* img) { return ::resolve_in(nn, buf, img, b0); };
    auto resolve_out = [&](romp_net* nn, int buf, float* c, float* p) { return ::resolve_out(nn, buf, c, p, b0); };
    switch (op.kind) {
        case ROMP_OP_STEM: {
            const float* in = resolve_in(n, op.in_buf, image);
            float* out = resolve_out(n, op.out_buf, center, params);
            ROMP_REQUIRE(in && out, "stem: bad buffers %d -> %d", op.in_buf, op.out_buf);
            return launch_stem(op, in, out, B, st);
        }
        case ROMP_OP_STEM7: {
            const float* in = resolve_in(n, op.in_buf, image);
            float* out = resolve_out(n, op.out_buf, center, params);
            ROMP_REQUIRE(in && out, "stem7: bad buffers %d -> %d", op.in_buf, op.out_buf);
            return launch_stem7(op, in, out, B, st);
        }
        case ROMP_OP_STEM7P: {
            float* out = resolve_out(n, op.out_buf, center, params);
            ROMP_REQUIRE(image && out && op.in_buf == ROMP_BUF_IMAGE, "stem7p: bad buffers");
            return launch_stem7p(op, image, out, B, st);
        }
        case ROMP_OP_MAXPOOL: {
            const float* in = resolve_in(n, op.in_buf, image);
            float* out = resolve_out(n, op.out_buf, center, params);
            ROMP_REQUIRE(in && out, "maxpool: bad buffers %d -> %d", op.in_buf, op.out_buf);
            return launch_maxpool(op, in, out, B, st);
        }
        case ROMP_OP_BEV_PACK: {
            const float* fv = resolve_in(n, op.in_buf, image);
            const float* ft = resolve_in(n, op.res_buf, image);
            float* out = resolve_out(n, op.out_buf, center, params);
            ROMP_REQUIRE(fv && ft && out, "bev_pack: bad buffers");
            return launch_bev_pack(fv, op.in_cstride, ft, op.res_cstride, out, B, st);
        }
        case ROMP_OP_BEV_MAPS: {
            const float* fv = resolve_in(n, op.in_buf, image);
            const float* bv = resolve_in(n, op.res_buf, image);
            float* c3 = resolve_out(n, op.out_buf, center, params);
            float* cam = resolve_out(n, op.term_buf[0], center, params);
            ROMP_REQUIRE(fv && bv && c3 && cam && op.weight, "bev_maps: bad buffers");
            return launch_bev_maps(fv, op.in_cstride, bv, op.res_cstride, op.weight, c3, cam, B, st);
        }
        case ROMP_OP_CONV3D: {
            const float* in = resolve_in(n, op.in_buf, image);
            const float* res = op.res_buf == ROMP_BUF_NONE ? nullptr : resolve_in(n, op.res_buf, image);
            float* out = resolve_out(n, op.out_buf, center, params);
            ROMP_REQUIRE(in && out && op.weight && op.scale && op.shift, "conv3d: bad buffers");
            return launch_conv3d(op.Cin, op.weight, op.scale, op.shift, op.relu, in, res, out, B, st);
        }
        case ROMP_OP_CONV: {
            if (op.ksize == 13) {
                // Conv1d k=3 along W over B independent sequences: one "image" whose rows are the batch
                const float* in1 = resolve_in(n, op.in_buf, image);
                float* out1 = resolve_out(n, op.out_buf, center, params);
                ROMP_REQUIRE(in1 && out1 && op.res_buf == ROMP_BUF_NONE, "conv1d: bad buffers");
                romp_op o1 = op;
                o1.H = B;
                return launch_conv(o1, in1, nullptr, out1, 1, n->mode, variant, queue, st, n->wg_cap);
            }
            const float* in = resolve_in(n, op.in_buf, image);
            float* out = resolve_out(n, op.out_buf, center, params);
            const float* res = op.res_buf == ROMP_BUF_NONE ? nullptr : resolve_in(n, op.res_buf, image);
            ROMP_REQUIRE(in && out, "conv: bad buffers %d -> %d", op.in_buf, op.out_buf);
            ROMP_REQUIRE(op.res_buf == ROMP_BUF_NONE || res, "conv: bad residual buffer %d", op.res_buf);
            return launch_conv(op, in, res, out, B, n->mode, variant, queue, st, n->wg_cap);
        }
        case ROMP_OP_FUSESUM: {
            FuseTerm t[4];
            ROMP_REQUIRE(op.n_terms >= 1 && op.n_terms <= 4, "fusesum: n_terms %d", op.n_terms);
            for (int k = 0; k < op.n_terms; ++k) {
                t[k].ptr = resolve_in(n, op.term_buf[k], image);
                ROMP_REQUIRE(t[k].ptr, "fusesum: bad term buffer %d", op.term_buf[k]);
                ROMP_REQUIRE(op.term_coff[k] >= 0 && (op.term_coff[k] & 7) == 0, "fusesum: term channel offset %d must be a multiple of 8", op.term_coff[k]);
                t[k].ptr += op.term_coff[k];
                t[k].shift = op.term_shift[k];
                t[k].cstride = op.term_cstride[k];
                t[k].fmt = op.term_fmt[k];
            }
            float* out = resolve_out(n, op.out_buf, center, params);
            ROMP_REQUIRE(out, "fusesum: bad out buffer %d", op.out_buf);
            return launch_fusesum(t, op.n_terms, out, B, op.H, op.W, op.Cout, op.out_cstride, op.out_coff, op.relu, st, op.out_fmt, op.act_shift);
        }
        case ROMP_OP_FUSEUP: {
            FuseTerm t[4];
            ROMP_REQUIRE(op.n_terms >= 2 && op.n_terms <= 4, "fuseup: n_terms %d", op.n_terms);
            for (int k = 0; k < op.n_terms; ++k) {
                t[k].ptr = resolve_in(n, op.term_buf[k], image);
                ROMP_REQUIRE(t[k].ptr && op.term_coff[k] >= 0, "fuseup: bad term buffer %d", op.term_buf[k]);
                t[k].ptr += op.term_coff[k];
                t[k].shift = op.term_shift[k]; t[k].cstride = op.term_cstride[k]; t[k].fmt = op.term_fmt[k];
            }
            float* out = resolve_out(n, op.out_buf, center, params);
            ROMP_REQUIRE(out, "fuseup: bad out buffer %d", op.out_buf);
            return launch_fuseup(op, t, out, B, st);
        }
        case ROMP_OP_KSUM: {
            const float* part = resolve_in(n, op.in_buf, image);
            const float* res = op.res_buf == ROMP_BUF_NONE ? nullptr : resolve_in(n, op.res_buf, image);
            float* out = resolve_out(n, op.out_buf, center, params);
            ROMP_REQUIRE(part && out && (op.res_buf == ROMP_BUF_NONE || res), "ksum: bad buffers %d -> %d", op.in_buf, op.out_buf);
            return launch_ksum(op, part, res, out, B, st);
        }
        case ROMP_OP_BBLOCK32: {
            ROMP_REQUIRE(idx > 0 && n->ops[idx - 1].kind == ROMP_OP_NOP, "bblock32: the op before it must be the NOP holding the first conv");
            const float* x = resolve_in(n, op.res_buf, image);
            float* y = resolve_out(n, op.out_buf, center, params);
            ROMP_REQUIRE(x && y && n->ops[idx - 1].in_buf == op.res_buf, "bblock32: bad buffers %d -> %d", op.res_buf, op.out_buf);
            return launch_bblock32(n->ops[idx - 1], op, x, y, B, queue, st);
        }
        case ROMP_OP_BBLOCK64: {
            ROMP_REQUIRE(idx > 0 && n->ops[idx - 1].kind == ROMP_OP_NOP, "bblock64: the op before it must be the NOP holding the first conv");
            const float* x = resolve_in(n, op.res_buf, image);
            float* y = resolve_out(n, op.out_buf, center, params);
            ROMP_REQUIRE(x && y && n->ops[idx - 1].in_buf == op.res_buf, "bblock64: bad buffers %d -> %d", op.res_buf, op.out_buf);
            return launch_bblock64(n->ops[idx - 1], op, x, y, B, queue, st);
        }
        case ROMP_OP_STEM2: {
            ROMP_REQUIRE(idx > 0 && n->ops[idx - 1].kind == ROMP_OP_NOP && n->ops[idx - 1].in_buf == ROMP_BUF_IMAGE, "stem2: the op before it must be the NOP holding the stem");
            float* y = resolve_out(n, op.out_buf, center, params);
            ROMP_REQUIRE(image && y, "stem2: bad buffers");
            return launch_stem2(n->ops[idx - 1], op, image, y, B, st);
        }
        case ROMP_OP_SEAM1X1: {
            ROMP_REQUIRE(idx > 0 && n->ops[idx - 1].kind == ROMP_OP_NOP, "seam1x1: the op before it must be the NOP holding the first conv");
            const romp_op& a = n->ops[idx - 1];
            // ROMP_OPF_SEAM_DS: the NOP before THAT one is the downsample conv that produced the residual; the kernel runs it too
            ROMP_REQUIRE(!(op.flags & ROMP_OPF_SEAM_DS) || idx >= 2, "seam1x1: ROMP_OPF_SEAM_DS needs two ops in front of it");
            const romp_op* d = (op.flags & ROMP_OPF_SEAM_DS) ? &n->ops[idx - 2] : nullptr;
            ROMP_REQUIRE(!d || d->kind == ROMP_OP_NOP, "seam1x1: the op two before it must be the NOP holding the downsample conv");
            const float* m = resolve_in(n, a.in_buf, image);
            const float* x = resolve_in(n, d ? d->in_buf : a.res_buf, image);
            float* t = resolve_out(n, a.out_buf, center, params);
            float* u = resolve_out(n, op.out_buf, center, params);
            ROMP_REQUIRE(m && x && t && u && op.in_buf == a.out_buf, "seam1x1: bad buffers");
            return launch_seam1x1(a, op, d, m, x, t, u, B, st);
        }
        case ROMP_OP_NOP:
        case ROMP_OP_FORK:
        case ROMP_OP_JOIN:
        case ROMP_OP_RECORD:
        case ROMP_OP_WAIT:
            return ROMP_OK;              // stream markers: handled by run_all; NOP: fused into the next op
        default:
            set_error("unknown op kind %d", op.kind);
            return ROMP_EINVAL;
    }
}

// Captured graphs hold the kernel choices / stream topology of the moment they were captured.  Dropping them is rare
// (re-tuning, option changes, cache overflow, teardown); the device is drained first so that no replay of a graph --
// nor any of the runtime's deferred clean-up for one -- is still in flight when its executable is destroyed.
static void drop_graphs(romp_net* n) {
    if (n->graphs.empty()) return;
    (void)hipDeviceSynchronize();
    for (auto& kv : n->graphs) hipGraphExecDestroy(kv.second);
    n->graphs.clear();
}

static const std::vector<int>* tuned_for(romp_net* n, int B) {
    auto it = n->tuned.find(B);
    return it == n->tuned.end() ? nullptr : &it->second;
}

static int reset_queues(romp_net* n, hipStream_t st) {
    ROMP_HIP_CHECK(hipMemsetAsync(n->queues, 0, 2 * n->ops.size() * QUEUE_INTS * sizeof(int), st));
    return ROMP_OK;
}

// One op sequence (a whole batch, or one batch lane) on `st` with its FORK/JOIN regions on the lane's side streams.
static int run_lane(romp_net* n, const float* image, int B, float* center, float* params, hipStream_t st, int lane, int b0,
                    hipEvent_t ev_after_first_convs, size_t first_op = 0) {
    const std::vector<int>* tv = tuned_for(n, B);
    const bool ms = n->use_streams && n->mode == 0;
    int convs_seen = 0;
    for (size_t i = first_op; i < n->ops.size(); ++i) {
        const romp_op& op = n->ops[i];
        if (op.kind == ROMP_OP_FORK) {
            if (!ms) continue;
            ROMP_REQUIRE(op.Cin >= 1 && op.Cin <= 3, "fork: %d side streams", op.Cin);
            ROMP_HIP_CHECK(hipEventRecord(n->ev_fork[lane], st));
            for (int k = 0; k < op.Cin; ++k) ROMP_HIP_CHECK(hipStreamWaitEvent(n->side[lane][k], n->ev_fork[lane], 0));
            continue;
        }
        if (op.kind == ROMP_OP_JOIN) {
            if (!ms) continue;
            for (int k = 0; k < op.Cin; ++k) {
                ROMP_HIP_CHECK(hipEventRecord(n->ev_join[lane][k], n->side[lane][k]));
                ROMP_HIP_CHECK(hipStreamWaitEvent(st, n->ev_join[lane][k], 0));
            }
            continue;
        }
        hipStream_t s = (ms && op.stream >= 1 && op.stream <= 3) ? n->side[lane][op.stream - 1] : st;
        if (op.kind == ROMP_OP_RECORD || op.kind == ROMP_OP_WAIT) {
            if (!ms) continue;
            ROMP_REQUIRE(op.Cin >= 0 && op.Cin < (int)n->ev_edge[lane].size(), "record / wait: event %d of %d", op.Cin, (int)n->ev_edge[lane].size());
            if (op.kind == ROMP_OP_RECORD) ROMP_HIP_CHECK(hipEventRecord(n->ev_edge[lane][op.Cin], s));
            else ROMP_HIP_CHECK(hipStreamWaitEvent(s, n->ev_edge[lane][op.Cin], 0));
            continue;
        }
        const int rc = run_op(n, i, tv ? (*tv)[i] : -1, image, B, center, params, s, lane, b0);
        if (rc) return rc;
        if (ev_after_first_convs && op.kind == ROMP_OP_CONV && ++convs_seen == 2)      // lane 1 starts ~two layers behind lane 0
            ROMP_HIP_CHECK(hipEventRecord(ev_after_first_convs, st));
    }
    return ROMP_OK;
}

// The forward's hipGraph, built EXPLICITLY (round 4).  Rounds 1-3 captured run_lane with its side streams; the stage region's
// point-to-point edges cannot be captured on this runtime: hipStreamWaitEvent during a capture appends every non-origin waiting stream
// to the recording stream's list of parallel capture streams, two side streams that wait for each other form a cycle, and
// hipStreamEndCapture walks those lists recursively -- off the end of the stack (ROCm 7.0 / 7.2; backtrace in profiles/r04_notes.md).
// So: every op is captured ALONE on a scratch stream (a one-stream capture: a chain of kernel / memset nodes), its nodes are re-added
// to the forward's graph, and the dependencies come from the program: stream order, FORK / JOIN, RECORD / WAIT.  `whole`: a whole
// forward (queue reset in front, as run_all) -- otherwise one batch lane (as run_lane; the caller resets the queues).
static int build_graph(romp_net* n, const float* image, int B, float* center, float* params, int lane, int b0, size_t first_op, bool whole,
                       hipGraph_t* out) {
    typedef std::vector<hipGraphNode_t> Nodes;
    struct Owned {                                             // a graph that dies with its scope unless released
        hipGraph_t g = nullptr;
        ~Owned() { if (g) hipGraphDestroy(g); }
    };
    Owned top;
    ROMP_HIP_CHECK(hipGraphCreate(&top.g, 0));
    hipGraph_t G = top.g;
    Nodes frontier[4];                                         // per stream: the nodes its next op depends on
    std::map<int, Nodes> events;
    auto merge = [](Nodes& into, const Nodes& from) {
        for (hipGraphNode_t x : from)
            if (std::find(into.begin(), into.end(), x) == into.end()) into.push_back(x);
    };
    auto add_op = [&](int s, auto&& body) -> int {             // capture `body` on the scratch stream, move its nodes behind frontier[s]
        Owned one;
        ROMP_HIP_CHECK(hipStreamBeginCapture(n->scratch, hipStreamCaptureModeThreadLocal));
        const int rc = body(n->scratch);
        const hipError_t e = hipStreamEndCapture(n->scratch, &one.g);
        if (rc) return rc;
        ROMP_HIP_CHECK(e);
        hipGraph_t g1 = one.g;
        size_t cnt = 0;
        ROMP_HIP_CHECK(hipGraphGetNodes(g1, nullptr, &cnt));
        if (cnt) {
            size_t n_root = 0;
            ROMP_HIP_CHECK(hipGraphGetRootNodes(g1, nullptr, &n_root));
            ROMP_REQUIRE(n_root == 1, "build_graph: an op captured as %zu root nodes (a chain is expected)", n_root);
            hipGraphNode_t cur = nullptr;
            ROMP_HIP_CHECK(hipGraphGetRootNodes(g1, &cur, &n_root));
            for (size_t k = 0; k < cnt; ++k) {
                hipGraphNodeType type;
                ROMP_HIP_CHECK(hipGraphNodeGetType(cur, &type));
                hipGraphNode_t nn = nullptr;
                if (type == hipGraphNodeTypeKernel) {
                    hipKernelNodeParams kp;
                    ROMP_HIP_CHECK(hipGraphKernelNodeGetParams(cur, &kp));
                    ROMP_HIP_CHECK(hipGraphAddKernelNode(&nn, G, frontier[s].data(), frontier[s].size(), &kp));
                } else if (type == hipGraphNodeTypeMemset) {
                    hipMemsetParams mp;
                    ROMP_HIP_CHECK(hipGraphMemsetNodeGetParams(cur, &mp));
                    ROMP_HIP_CHECK(hipGraphAddMemsetNode(&nn, G, frontier[s].data(), frontier[s].size(), &mp));
                } else {
                    set_error("build_graph: an op captured a node of type %d", (int)type);
                    return ROMP_EHIP;
                }
                frontier[s].assign(1, nn);
                if (k + 1 < cnt) {
                    size_t n_next = 0;
                    ROMP_HIP_CHECK(hipGraphNodeGetDependentNodes(cur, nullptr, &n_next));
                    ROMP_REQUIRE(n_next == 1, "build_graph: a captured node has %zu dependents (a chain is expected)", n_next);
                    ROMP_HIP_CHECK(hipGraphNodeGetDependentNodes(cur, &cur, &n_next));
                }
            }
        }
        return ROMP_OK;
    };
    conv_set_sat_counter(n->sat, n->sat_checked);
    const int cap = n->wg_cap;
    if (whole && n->split == 2) n->wg_cap = 0;                 // an unsplit forward takes the whole chip (run_all)
    conv_set_wg_cap(n->wg_cap);
    int rc = whole ? add_op(0, [&](hipStream_t s) { return reset_queues(n, s); }) : ROMP_OK;
    const std::vector<int>* tv = tuned_for(n, B);
    const bool ms = n->use_streams && n->mode == 0;
    for (size_t i = first_op; i < n->ops.size() && !rc; ++i) {
        const romp_op& op = n->ops[i];
        const int s = (ms && op.stream >= 1 && op.stream <= 3) ? op.stream : 0;
        switch (op.kind) {
            case ROMP_OP_FORK:
                for (int k = 1; ms && k <= op.Cin && k <= 3; ++k) merge(frontier[k], frontier[0]);
                break;
            case ROMP_OP_JOIN:
                for (int k = 1; ms && k <= op.Cin && k <= 3; ++k) merge(frontier[0], frontier[k]);
                break;
            case ROMP_OP_RECORD:
                if (ms) events[op.Cin] = frontier[s];
                break;
            case ROMP_OP_WAIT:
                if (ms) merge(frontier[s], events[op.Cin]);
                break;
            case ROMP_OP_NOP:
                break;
            default:
                rc = add_op(s, [&](hipStream_t st) { return run_op(n, i, tv ? (*tv)[i] : -1, image, B, center, params, st, lane, b0); });
        }
    }
    n->wg_cap = cap;
    if (rc) return rc;
    *out = top.g;
    top.g = nullptr;
    return ROMP_OK;
}

static bool lanes_active(const romp_net* n, int B) { return n->split == 2 && n->mode == 0 && B >= 2 && !(B & 1); }

static int run_all(romp_net* n, const float* image, int B, float* center, float* params, hipStream_t st, size_t first_op = 0) {
    conv_set_sat_counter(n->sat, n->sat_checked);
    conv_set_wg_cap(n->wg_cap);
    int rc = reset_queues(n, st);
    if (rc) return rc;
    if (!lanes_active(n, B)) {
        const int cap = n->wg_cap;
        if (n->split == 2) n->wg_cap = 0;              // an unsplit forward (odd / single-image batch) takes the whole chip
        rc = run_lane(n, image, B, center, params, st, 0, 0, nullptr, first_op);
        n->wg_cap = cap;
        return rc;
    }
    const int B0 = B / 2;
    ROMP_HIP_CHECK(hipEventRecord(n->ev_begin, st));                        // after the queue reset
    ROMP_HIP_CHECK(hipEventRecord(n->ev_offset, st));                       // (re-recorded two convs into lane 0)
    rc = run_lane(n, image, B0, center, params, st, 0, 0, n->ev_offset);
    if (rc) return rc;
    ROMP_HIP_CHECK(hipStreamWaitEvent(n->lane_main, n->ev_begin, 0));
    ROMP_HIP_CHECK(hipStreamWaitEvent(n->lane_main, n->ev_offset, 0));
    rc = run_lane(n, image + (size_t)B0 * n->image_floats, B0, center + (size_t)B0 * n->center_floats,
                  params + (size_t)B0 * n->params_floats, n->lane_main, 1, B0, nullptr);
    if (rc) return rc;
    ROMP_HIP_CHECK(hipEventRecord(n->ev_done, n->lane_main));
    ROMP_HIP_CHECK(hipStreamWaitEvent(st, n->ev_done, 0));
    return ROMP_OK;
}

extern "C" {

int romp_abi_version(void) { return ROMP_ABI_VERSION; }
const char* romp_last_error(void) { return romp::g_err; }

int romp_net_create(romp_net** out, const romp_op* ops_host, int n_ops, const int64_t* buf_floats, int n_bufs,
                    int max_batch) {
    ROMP_REQUIRE(out && ops_host && n_ops > 0 && n_bufs >= 0 && max_batch > 0, "romp_net_create: bad arguments");
    {   // One device per process (the design: one process per GPU, torch.distributed over RCCL): the launchers cache their one-time
        // set-up -- raised dynamic-LDS limits, CU count, zero pages, occupancy -- per PROCESS.  A second net on another device would
        // silently run with the first device's, so it is refused here.
        static int g_device = -1;
        int dev = -1;
        ROMP_HIP_CHECK(hipGetDevice(&dev));
        if (g_device < 0) g_device = dev;
        ROMP_REQUIRE(dev == g_device, "romp_net_create: this process already runs nets on device %d; libromp_hip.so serves ONE device per "
                     "process (launch one process per GPU), current device is %d", g_device, dev);
    }
    { const int rc = conv_init(); if (rc) return rc; }
    for (int i = 1; i < n_ops; ++i)
        if (ops_host[i].kind == ROMP_OP_SEAM1X1) {
            const int rc = launch_seam1x1(ops_host[i - 1], ops_host[i], nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr);
            if (rc) return rc;
            break;
        }
    for (int i = 1; i < n_ops; ++i)
        if (ops_host[i].kind == ROMP_OP_STEM2) {
            const int rc = launch_stem2(ops_host[i - 1], ops_host[i], nullptr, nullptr, 0, nullptr);
            if (rc) return rc;
            break;
        }
    for (int i = 0; i < n_ops; ++i)
        if (ops_host[i].kind == ROMP_OP_STEM7P) {
            const int rc = launch_stem7p(ops_host[i], nullptr, nullptr, 0, nullptr);
            if (rc) return rc;
            break;
        }
    for (int i = 0; i < n_ops; ++i)                            // (every instantiation in use: cheap, idempotent)
        if (ops_host[i].kind == ROMP_OP_FUSEUP) {
            const int rc = launch_fuseup(ops_host[i], nullptr, nullptr, 0, nullptr);
            if (rc) return rc;
        }
    for (int i = 0, seen = 0; i < n_ops && seen != 3; ++i) {  // the fused blocks' one-time set-up (hipMalloc / attributes) must not run inside a stream capture
        const int kind = ops_host[i].kind;
        if (kind == ROMP_OP_BBLOCK32 && !(seen & 1) && i > 0) {
            const int rc = launch_bblock32(ops_host[i - 1], ops_host[i], nullptr, nullptr, 0, nullptr, nullptr);
            if (rc) return rc;
            seen |= 1;
        }
        if (kind == ROMP_OP_BBLOCK64 && !(seen & 2) && i > 0) {
            const int rc = launch_bblock64(ops_host[i - 1], ops_host[i], nullptr, nullptr, 0, nullptr, nullptr);
            if (rc) return rc;
            seen |= 2;
        }
    }
    romp_net* n = new romp_net();
    n->ops.assign(ops_host, ops_host + n_ops);
    n->buf_floats.assign(buf_floats, buf_floats + n_bufs);
    n->max_batch = max_batch;
    // graph replay launches op 0 eagerly and bakes no image pointer into the graph: only sound if nothing else reads the image
    n->image_only_in_op0 = n_ops > 0 && (n->ops[0].kind == ROMP_OP_STEM || n->ops[0].kind == ROMP_OP_STEM7);
    if (n_ops > 1 && n->ops[0].kind == ROMP_OP_NOP && n->ops[0].in_buf == ROMP_BUF_IMAGE && (n->ops[1].kind == ROMP_OP_STEM2 || n->ops[1].kind == ROMP_OP_STEM7P)) {
        n->image_only_in_op0 = true;                           // the fused stem: the NOP that holds the stem's fields + the kernel's op
        n->eager_ops = 2;
    }
    for (int i = n->eager_ops; i < n_ops && n->image_only_in_op0; ++i) {
        const romp_op& o = n->ops[i];
        bool reads = o.in_buf == ROMP_BUF_IMAGE || o.res_buf == ROMP_BUF_IMAGE;
        for (int k = 0; k < 4; ++k) reads |= ((o.kind == ROMP_OP_FUSESUM || o.kind == ROMP_OP_FUSEUP) && k < o.n_terms && o.term_buf[k] == ROMP_BUF_IMAGE);
        if (reads) n->image_only_in_op0 = false;
    }
    // a flag bit this build does not know changes what an op MEANS (ROMP_OPF_SEAM_DS: the residual tensor is never written): a
    // program lowered for a newer ABI must fail here, not run as something else
    for (int i = 0; i < n_ops; ++i)
        if (n->ops[i].flags & ~ROMP_OPF_ALL) {
            set_error("op %d: unknown flag bits 0x%x (library ABI %d)", i, n->ops[i].flags & ~ROMP_OPF_ALL, ROMP_ABI_VERSION);
            romp_net_destroy(n);
            return ROMP_EINVAL;
        }
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
    if (hipMalloc((void**)&n->queues, (size_t)2 * n_ops * QUEUE_INTS * sizeof(int)) != hipSuccess ||
        hipMalloc((void**)&n->sat, 64) != hipSuccess || hipMemset(n->sat, 0, 64) != hipSuccess) {
        set_error("queue allocation failed");
        romp_net_destroy(n);
        return ROMP_ENOMEM;
    }
    { const char* e = getenv("ROMP_CHECK_FINITE"); n->sat_checked = e && e[0] && strcmp(e, "0") != 0; }
    bool ok = hipStreamCreateWithFlags(&n->lane_main, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&n->scratch, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&n->ev_begin, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&n->ev_offset, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&n->ev_done, hipEventDisableTiming) == hipSuccess;
    for (int l = 0; l < 2 && ok; ++l) {
        ok = hipEventCreateWithFlags(&n->ev_fork[l], hipEventDisableTiming) == hipSuccess;
        for (int k = 0; k < 3 && ok; ++k)
            ok = hipStreamCreateWithFlags(&n->side[l][k], hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&n->ev_join[l][k], hipEventDisableTiming) == hipSuccess;
    }
    int n_edge = 0;
    {
        std::vector<char> recorded;
        for (const romp_op& op : n->ops) {
            if (op.kind != ROMP_OP_RECORD && op.kind != ROMP_OP_WAIT) continue;
            bool good = op.Cin >= 0 && op.Cin < 4096 && op.stream >= 0 && op.stream <= 3;
            if (good && op.kind == ROMP_OP_RECORD) { if ((int)recorded.size() <= op.Cin) recorded.resize(op.Cin + 1, 0); recorded[op.Cin] = 1; }
            if (good && op.kind == ROMP_OP_WAIT) good = op.Cin < (int)recorded.size() && recorded[op.Cin];       // op order is a valid serial order
            if (!good) { set_error("record / wait op: event %d on stream %d (a wait needs an earlier record)", op.Cin, op.stream); romp_net_destroy(n); return ROMP_EINVAL; }
            n_edge = std::max(n_edge, op.Cin + 1);
        }
    }
    for (int l = 0; l < 2 && ok && n_edge > 0 && n_edge <= 4096; ++l) {
        n->ev_edge[l].assign(n_edge, nullptr);
        for (int k = 0; k < n_edge && ok; ++k) ok = hipEventCreateWithFlags(&n->ev_edge[l][k], hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
        set_error("side stream / event creation failed");
        romp_net_destroy(n);
        return ROMP_EHIP;
    }
    *out = n;
    return ROMP_OK;
}

int romp_net_set_mode(romp_net* n, int mode) {
    ROMP_REQUIRE(n && (mode == 0 || mode == 1), "romp_net_set_mode: bad arguments");
    n->mode = mode;
    return ROMP_OK;
}

int romp_net_set_streams(romp_net* n, int enable) {
    ROMP_REQUIRE(n, "romp_net_set_streams: null net");
    n->use_streams = enable ? 1 : 0;
    drop_graphs(n);
    return ROMP_OK;
}

int romp_net_set_split(romp_net* n, int lanes, int wg_cap, int64_t image_floats, int64_t center_floats, int64_t params_floats) {
    ROMP_REQUIRE(n && (lanes == 1 || lanes == 2) && wg_cap >= 0, "romp_net_set_split: bad arguments");
    ROMP_REQUIRE(lanes == 1 || (image_floats > 0 && center_floats > 0 && params_floats > 0), "romp_net_set_split: per-image sizes missing");
    n->split = lanes;
    n->wg_cap = wg_cap;
    n->image_floats = image_floats; n->center_floats = center_floats; n->params_floats = params_floats;
    drop_graphs(n);
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
    conv_set_sat_counter(n->sat, n->sat_checked);
    // Streams off (the profiling configuration: every op on the caller's stream) replays EAGERLY even in graph mode.  A forward
    // without branch streams is a single-chain hipGraph, and ROCm 7.x's packet-capture path for such graphs faults (GPU memory
    // access fault, 3 of 6 runs) when the host calls hipDeviceSynchronize between replays of a short job -- gone with
    // DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, with a stream synchronisation instead, without the graph, and with the multi-branch graphs
    // every other configuration builds (profiles/r06_serial_graph_fault.txt: the bisect).  Not ours to fix; not worth a graph.
    if (!n->use_graph || n->mode != 0 || !n->use_streams) return run_all(n, image, B, center, params, st);
    ROMP_REQUIRE(st != nullptr, "graph mode needs a non-default stream");
    // The stem is the only op that reads the caller's image: launched eagerly in front of the graph, the graph no longer
    // depends on WHERE the input lives (a caller streaming frames from ever new tensors replays one graph).
    const bool stem_out = !lanes_active(n, B) && n->image_only_in_op0;
    const float* key_image = stem_out ? nullptr : image;
    if (n->graphs.size() >= 32 && !n->graphs.count(GraphKey{B, key_image, center, params, lanes_active(n, B) ? 0 : -1})) {
        // a caller that hands over new tensors every call must not grow the cache for ever
        ROMP_HIP_CHECK(hipStreamSynchronize(st));
        ROMP_HIP_CHECK(hipStreamSynchronize(n->lane_main));
        drop_graphs(n);
    }
    auto build = [&](const GraphKey& key, const float* im, int Bl, float* ce, float* pa, int lane, int b0, size_t first_op, bool whole) -> int {
        if (n->graphs.count(key)) return ROMP_OK;
        hipGraph_t g = nullptr;
        const int rc = build_graph(n, im, Bl, ce, pa, lane, b0, first_op, whole, &g);
        if (rc) return rc;
        hipGraphExec_t ge = nullptr;
        const hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
        ROMP_HIP_CHECK(e);
        n->graphs.emplace(key, ge);
        return ROMP_OK;
    };
    if (!lanes_active(n, B)) {
        const GraphKey key{B, key_image, center, params, -1};
        if (stem_out) {
            const int rc0 = run_op(n, n->eager_ops - 1, -1, image, B, center, params, st);     // (the ops in front of it in the prefix are NOPs)
            if (rc0) return rc0;
        }
        const int rc = build(key, image, B, center, params, 0, 0, stem_out ? n->eager_ops : 0, true);
        if (rc) return rc;
        ROMP_HIP_CHECK(hipGraphLaunch(n->graphs[key], st));
        return ROMP_OK;
    }
    // Batch lanes: one graph per lane, replayed on the lane's own stream.
    const int B0 = B / 2;
    const float* image1 = image + (size_t)B0 * n->image_floats;
    float* center1 = center + (size_t)B0 * n->center_floats;
    float* params1 = params + (size_t)B0 * n->params_floats;
    const GraphKey k0{B, image, center, params, 0}, k1{B, image, center, params, 1};
    int rc = build(k0, image, B0, center, params, 0, 0, 0, false);
    if (rc) return rc;
    rc = build(k1, image1, B0, center1, params1, 1, B0, 0, false);
    if (rc) return rc;
    rc = reset_queues(n, st);
    if (rc) return rc;
    ROMP_HIP_CHECK(hipEventRecord(n->ev_begin, st));
    ROMP_HIP_CHECK(hipStreamWaitEvent(n->lane_main, n->ev_begin, 0));
    ROMP_HIP_CHECK(hipGraphLaunch(n->graphs[k0], st));
    ROMP_HIP_CHECK(hipGraphLaunch(n->graphs[k1], n->lane_main));
    ROMP_HIP_CHECK(hipEventRecord(n->ev_done, n->lane_main));
    ROMP_HIP_CHECK(hipStreamWaitEvent(st, n->ev_done, 0));
    return ROMP_OK;
}

// Measure every valid kernel variant of every conv op at batch B (on the arena's own buffers: the
// data is whatever the last forward left there, timing does not depend on it) and keep the fastest.
int romp_net_autotune(romp_net* n, int B, int iters, void* stream) {
    ROMP_REQUIRE(n && B > 0 && iters > 0, "romp_net_autotune: bad arguments");
    if (B > n->max_batch) { set_error("batch %d > max_batch %d", B, n->max_batch); return ROMP_ECAPACITY; }
    hipStream_t st = (hipStream_t)stream;
    conv_set_sat_counter(nullptr, false);             // timing runs on whatever the arena holds: nothing to report
    std::vector<int> best(n->ops.size(), -1);
    // Scratch target for ops that read the caller's image or write the caller's output tensors: sized from the program
    // (largest per-image extent any conv touches through a pseudo buffer), not from one model's head.
    size_t scratch_floats = 16;
    for (const romp_op& op : n->ops) {
        if (op.kind != ROMP_OP_CONV) continue;
        const int kh = op.ksize == 13 ? 1 : op.ksize, kw = op.ksize == 13 ? 3 : op.ksize;
        const int Ho = op.ksize == 2 ? op.H / op.stride : (op.H + 2 * (kh / 2) - kh) / op.stride + 1;
        const int Wo = op.ksize == 2 ? op.W / op.stride : (op.W + 2 * (kw / 2) - kw) / op.stride + 1;
        if (op.in_buf == ROMP_BUF_IMAGE) scratch_floats = std::max(scratch_floats, (size_t)op.H * op.W * op.in_cstride);
        if (op.res_buf == ROMP_BUF_IMAGE) scratch_floats = std::max(scratch_floats, (size_t)Ho * Wo * op.res_cstride);
        if (op.out_buf == ROMP_BUF_CENTER || op.out_buf == ROMP_BUF_PARAMS)
            scratch_floats = std::max(scratch_floats, op.out_bstride > 0 ? (size_t)op.out_bstride : (size_t)Ho * Wo * op.out_cstride);
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float* scratch = nullptr;
    auto cleanup = [&]() {
        if (scratch) hipFree(scratch);
        if (e0) hipEventDestroy(e0);
        if (e1) hipEventDestroy(e1);
    };
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess ||
        hipMalloc((void**)&scratch, scratch_floats * (size_t)B * sizeof(float)) != hipSuccess) {
        cleanup();
        set_error("autotune: event / scratch allocation failed (%zu floats x %d)", scratch_floats, B);
        return ROMP_ENOMEM;
    }
    int rc = ROMP_OK;
    for (size_t i = 0; i < n->ops.size() && rc == ROMP_OK; ++i) {
        const romp_op& op = n->ops[i];
        if (op.kind != ROMP_OP_CONV) continue;
        { static const char* only = getenv("ROMP_AUTOTUNE_ONLY_OP");                 // debugging aid: measure one op only
          if (only && atoi(only) != (int)i) continue; }
        float best_ms = 1e30f;
        static const bool verbose = getenv("ROMP_AUTOTUNE_VERBOSE") != nullptr;      // debugging aid: names the launch a GPU fault belongs to
        for (int v = 0; v < conv_num_variants(); ++v) {
            if (!conv_variant_tunable(op, v)) continue;
            if (verbose) {
                char name[128] = "?";
                describe_conv(op, B, v, name, sizeof(name));
                fprintf(stderr, "autotune: op %zu (H %d Cin %d Cout %d k%d s%d g%d) variant %d %s\n", i, op.H, op.Cin, op.Cout, op.ksize, op.stride, op.groups, v, name);
                fflush(stderr);
            }
            float ms_min = 1e30f;
            for (int it = 0; it < iters + 1 && rc == ROMP_OK; ++it) {
                rc = reset_queues(n, st);
                if (rc) break;
                hipEventRecord(e0, st);
                rc = run_op(n, i, v, scratch, B, scratch, scratch, st);
                hipEventRecord(e1, st);
                if (hipEventSynchronize(e1) != hipSuccess) { set_error("autotune: kernel failed (op %zu variant %d)", i, v); rc = ROMP_EHIP; break; }
                float ms = 0.f;
                hipEventElapsedTime(&ms, e0, e1);
                if (it > 0 && ms < ms_min) ms_min = ms;      // first run is a warm-up
            }
            if (ms_min < best_ms) { best_ms = ms_min; best[i] = v; }
        }
    }
    cleanup();
    if (rc == ROMP_OK) {
        n->tuned[B] = best;
        drop_graphs(n);
    }
    return rc;
}

int romp_net_profile(romp_net* n, const float* image, int B, float* center, float* params, void* stream,
                     float* ms_out, int iters) {
    ROMP_REQUIRE(n && image && center && params && ms_out && B > 0 && iters > 0, "romp_net_profile: bad arguments");
    if (B > n->max_batch) { set_error("batch %d > max_batch %d", B, n->max_batch); return ROMP_ECAPACITY; }
    hipStream_t st = (hipStream_t)stream;
    conv_set_sat_counter(n->sat, n->sat_checked);
    const size_t nops = n->ops.size();
    std::vector<hipEvent_t> ev(nops + 1);
    for (auto& e : ev) ROMP_HIP_CHECK(hipEventCreate(&e));
    std::vector<std::vector<float>> samples(nops);
    const std::vector<int>* tv = tuned_for(n, B);
    int rc = ROMP_OK;
    for (int it = 0; it < iters && rc == ROMP_OK; ++it) {
        rc = reset_queues(n, st);
        if (rc) break;
        hipEventRecord(ev[0], st);
        for (size_t i = 0; i < nops; ++i) {
            rc = run_op(n, i, tv ? (*tv)[i] : -1, image, B, center, params, st);
            if (rc) break;
            hipEventRecord(ev[i + 1], st);
        }
        if (rc) break;
        if (hipStreamSynchronize(st) != hipSuccess) { set_error("hipStreamSynchronize failed"); rc = ROMP_EHIP; break; }
        for (size_t i = 0; i < nops; ++i) {
            float ms = 0.f;
            hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            samples[i].push_back(ms);
        }
    }
    for (auto& e : ev) hipEventDestroy(e);
    for (size_t i = 0; i < nops; ++i) {          // median of the passes: one pre-empted launch must not skew a class average
        std::vector<float>& v = samples[i];
        if (v.empty()) { ms_out[i] = 0.f; continue; }
        std::sort(v.begin(), v.end());
        ms_out[i] = v.size() & 1 ? v[v.size() / 2] : 0.5f * (v[v.size() / 2 - 1] + v[v.size() / 2]);
    }
    return rc;
}

// ---- activation range scan -----------------------------------------------------------------------------------------
// max |x| and the number of non-finite values of ONE op's output region: `C` channels at offset `coff` of every pixel (channel
// stride `cs`) of B images `bstride` floats apart.  H2 regions are decoded (h1 + h2) * 2^-shift per channel octet (an octet = 32
// bytes: eight high then eight low fp16 pieces).  One atomicMax on the float's bit pattern per wave.  (Round 3 reduced over the
// whole arena buffer: stale data of earlier live ranges and neighbouring channel slices made the figure an upper bound that could
// mask the too-small-for-fp16 test -- ADVICE r3.)
struct ScanRegion { const float* x; int B; long long bstride; long long npix; int cs, coff, C, h2; float inv_scale; };
__global__ void range_scan_kernel(ScanRegion r, unsigned* out_max, unsigned* out_bad) {
    float m = 0.f;
    unsigned bad = 0;
    const int per_pix = r.h2 ? r.C >> 3 : r.C;                 // work units per pixel: octets / single channels
    const size_t total = (size_t)r.B * r.npix * per_pix;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int u = (int)(i % per_pix);
        const size_t pixb = i / per_pix;
        const size_t pix = pixb % r.npix, b = pixb / r.npix;
        const float* px = r.x + b * r.bstride + pix * r.cs + r.coff;
        if (r.h2) {
            const uint4 hi = *reinterpret_cast<const uint4*>(px + u * 8), lo = *reinterpret_cast<const uint4*>(px + u * 8 + 4);
            const float4 a = h2_unpack(make_uint2(hi.x, hi.y), make_uint2(lo.x, lo.y), r.inv_scale);
            const float4 c = h2_unpack(make_uint2(hi.z, hi.w), make_uint2(lo.z, lo.w), r.inv_scale);
            const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (__builtin_isfinite(v[e])) m = fmaxf(m, fabsf(v[e])); else ++bad;
            }
        } else {
            const float v = px[u];
            if (__builtin_isfinite(v)) m = fmaxf(m, fabsf(v)); else ++bad;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        m = fmaxf(m, __shfl_xor(m, off));
        bad += __shfl_xor(bad, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(out_max, __float_as_uint(m));                // m >= 0: the bit patterns order like the values
        if (bad) atomicAdd(out_bad, bad);
    }
}

// the region an op writes into arena buffer op.out_buf: false for ops whose output is not a dense (pixels x channel slice) block
// (sparse parity outputs of a transposed conv, the BEV head's 3-D tensors): those are scanned as whole buffers
static bool op_out_region(const romp_op& op, long long* npix, int* cs, int* coff, int* C) {
    int Ho = op.H, Wo = op.W;
    switch (op.kind) {
        case ROMP_OP_CONV: {
            if (op.ksize == 13 || op.out_rstride > 0 || op.out_bstride > 0) return false;
            const int k = op.ksize;
            if (k == 2) { Ho = op.H / op.stride; Wo = op.W / op.stride; }
            else { Ho = (op.H + 2 * (k / 2) - k) / op.stride + 1; Wo = (op.W + 2 * (k / 2) - k) / op.stride + 1; }
            *C = op.groups > 1 ? (op.groups - 1) * op.out_gstride + op.Cout : op.Cout;
            break; }
        case ROMP_OP_STEM: case ROMP_OP_STEM7: Ho = op.H / 2; Wo = op.W / 2; *C = op.Cout; break;
        case ROMP_OP_FUSESUM: case ROMP_OP_FUSEUP: case ROMP_OP_KSUM: case ROMP_OP_BBLOCK32: case ROMP_OP_BBLOCK64: case ROMP_OP_SEAM1X1: *C = op.Cout; break;
        case ROMP_OP_STEM2: Ho = op.H / 2; Wo = op.W / 2; *C = op.Cout; break;      // (H x W of the op: conv2's input size)
        case ROMP_OP_STEM7P: Ho = op.H / 4; Wo = op.W / 4; *C = op.Cout; break;     // (H x W of the op: the image)
        default: return false;
    }
    *npix = (long long)Ho * Wo; *cs = op.out_cstride; *coff = op.out_coff;
    return *cs > 0 && *C > 0 && (op.out_fmt != ROMP_FMT_H2 || ((*C | *cs | *coff) & 7) == 0);
}

// Runs the program op by op (one stream, the variants of romp_net_autotune if it ran) and reports for every op that writes an
// arena buffer the max |x| / non-finite count of the region it wrote (B images) right after the op, and how many saturation events
// (conv_common.h sat_report: waves that clamped a value at +-65504 while splitting it into fp16 pieces) the op's kernels reported --
// the fused-block kernels run their counting builds here.  plan.assign_formats uses max |x| (of a float32 lowering) to keep tensors
// whose range does not fit the fp16 pieces of the H2 format in float32 and their consumers on the f32 / bf16x3 kernels; on the real
// program the saturation column says which op clamped (RompNet.range_scan).
int romp_net_range_scan(romp_net* n, const float* image, int B, float* center, float* params, void* stream,
                        float* maxabs_out_host, int32_t* nonfinite_out_host, int32_t* saturated_out_host) {
    ROMP_REQUIRE(n && image && center && params && maxabs_out_host && nonfinite_out_host && B > 0, "romp_net_range_scan: bad arguments");
    if (B > n->max_batch) { set_error("batch %d > max_batch %d", B, n->max_batch); return ROMP_ECAPACITY; }
    hipStream_t st = (hipStream_t)stream;
    const size_t nops = n->ops.size();
    unsigned* d = nullptr;                                     // [max | bad | counter after op i] x nops
    ROMP_HIP_CHECK(hipMalloc((void**)&d, 3 * nops * sizeof(unsigned)));
    int rc = ROMP_OK;
    if (hipMemsetAsync(d, 0, 3 * nops * sizeof(unsigned), st) != hipSuccess) { set_error("hipMemsetAsync failed"); rc = ROMP_EHIP; }
    int sat_before = 0;
    if (rc == ROMP_OK && (hipMemcpyAsync(&sat_before, n->sat, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) {
        set_error("range scan: reading the saturation counter failed"); rc = ROMP_EHIP;
    }
    // the kernels the FORWARD of this batch runs: with two lanes active every kernel sees B / 2 images and the table measured for that
    const std::vector<int>* tv = tuned_for(n, lanes_active(n, B) ? B / 2 : B);
    conv_set_sat_counter(n->sat, true);
    if (rc == ROMP_OK) rc = reset_queues(n, st);
    for (size_t i = 0; i < nops && rc == ROMP_OK; ++i) {
        rc = run_op(n, i, tv ? (*tv)[i] : -1, image, B, center, params, st);
        const romp_op& op = n->ops[i];
        if (rc == ROMP_OK && hipMemcpyAsync(d + 2 * nops + i, n->sat, sizeof(int), hipMemcpyDeviceToDevice, st) != hipSuccess) { set_error("hipMemcpyAsync failed"); rc = ROMP_EHIP; }
        if (rc || op.out_buf < 0 || op.out_buf >= (int)n->bufs.size() || op.kind == ROMP_OP_FORK || op.kind == ROMP_OP_JOIN || op.kind == ROMP_OP_NOP ||
            op.kind == ROMP_OP_RECORD || op.kind == ROMP_OP_WAIT) continue;
        ScanRegion r;
        r.x = n->bufs[op.out_buf]; r.B = B; r.bstride = n->buf_floats[op.out_buf];
        r.h2 = op.out_fmt == ROMP_FMT_H2; r.inv_scale = ldexpf(1.f, -op.act_shift);
        if (!op_out_region(op, &r.npix, &r.cs, &r.coff, &r.C)) {      // the whole buffer as one "pixel" per 8 floats / per float
            r.cs = r.h2 ? 8 : 1; r.coff = 0; r.C = r.cs; r.npix = r.bstride / r.cs;
        }
        hipLaunchKernelGGL(range_scan_kernel, dim3(1024), dim3(256), 0, st, r, d + i, d + nops + i);
        if (hipGetLastError() != hipSuccess) { set_error("range_scan_kernel launch failed"); rc = ROMP_EHIP; }
    }
    conv_set_sat_counter(n->sat, n->sat_checked);
    std::vector<unsigned> h(3 * nops, 0u);
    if (rc == ROMP_OK && hipMemcpyAsync(h.data(), d, 3 * nops * sizeof(unsigned), hipMemcpyDeviceToHost, st) != hipSuccess) { set_error("hipMemcpyAsync failed"); rc = ROMP_EHIP; }
    if (hipStreamSynchronize(st) != hipSuccess && rc == ROMP_OK) { set_error("hipStreamSynchronize failed"); rc = ROMP_EHIP; }
    hipFree(d);
    int prev = sat_before;
    for (size_t i = 0; i < nops; ++i) {
        float f;
        memcpy(&f, &h[i], sizeof(float));
        maxabs_out_host[i] = f;
        nonfinite_out_host[i] = (int32_t)h[nops + i];
        const int now = (int)h[2 * nops + i];
        if (saturated_out_host) saturated_out_host[i] = rc == ROMP_OK ? now - prev : 0;
        prev = now;
    }
    return rc;
}

// The net's saturation counter: the number of (wave, work item) events since create / the last reset in which a kernel clamped a
// value at +-65504 while splitting it into fp16 pieces of x * 2^act_shift -- an activation outside the calibrated range: the
// result is finite but wrong there.  0 for a healthy net.  Synchronises `stream`.
int romp_net_saturated(romp_net* n, int64_t* count_host, int reset, void* stream) {
    ROMP_REQUIRE(n && count_host, "romp_net_saturated: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    int v = 0;
    ROMP_HIP_CHECK(hipMemcpyAsync(&v, n->sat, sizeof(int), hipMemcpyDeviceToHost, st));
    if (reset) ROMP_HIP_CHECK(hipMemsetAsync(n->sat, 0, sizeof(int), st));
    ROMP_HIP_CHECK(hipStreamSynchronize(st));
    *count_host = v;
    return ROMP_OK;
}

// Device address of the counter: romp_parse_watch reads it back with the detection counts (the API's default-on range guard).
const int32_t* romp_net_sat_counter(romp_net* n) { return n ? (const int32_t*)n->sat : nullptr; }

// 1: the fused BasicBlock kernels run their counting builds too (default: env ROMP_CHECK_FINITE=1; the API's range guard turns it on).
int romp_net_set_sat_check(romp_net* n, int enable) {
    ROMP_REQUIRE(n, "romp_net_set_sat_check: null net");
    n->sat_checked = enable != 0;
    drop_graphs(n);
    return ROMP_OK;
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
    drop_graphs(n);
    for (float* p : n->bufs)
        if (p) hipFree(p);
    if (n->queues) hipFree(n->queues);
    if (n->sat) {
        if (conv_sat_counter() == n->sat) conv_set_sat_counter(nullptr, false);      // no launcher may keep a pointer into a dead net
        hipFree(n->sat);
    }
    if (n->plan_dev) hipFree(n->plan_dev);
    for (int l = 0; l < 2; ++l) {
        for (int k = 0; k < 3; ++k) {
            if (n->side[l][k]) hipStreamDestroy(n->side[l][k]);
            if (n->ev_join[l][k]) hipEventDestroy(n->ev_join[l][k]);
        }
        if (n->ev_fork[l]) hipEventDestroy(n->ev_fork[l]);
        for (hipEvent_t e : n->ev_edge[l]) if (e) hipEventDestroy(e);
    }
    if (n->lane_main) hipStreamDestroy(n->lane_main);
    if (n->scratch) hipStreamDestroy(n->scratch);
    if (n->ev_begin) hipEventDestroy(n->ev_begin);
    if (n->ev_offset) hipEventDestroy(n->ev_offset);
    if (n->ev_done) hipEventDestroy(n->ev_done);
    delete n;
}

int romp_conv_forward(const romp_op* op, const float* in, const float* res, float* out, int B, int mode, int variant,
                      void* stream) {
    ROMP_REQUIRE(op && in && out && B > 0, "romp_conv_forward: bad arguments");
    conv_set_sat_counter(nullptr, false);              // a stand-alone layer belongs to no net: nothing to report to
    if (op->kind == ROMP_OP_STEM) return launch_stem(*op, in, out, B, (hipStream_t)stream);
    ROMP_REQUIRE(op->kind == ROMP_OP_CONV, "romp_conv_forward: op kind %d", op->kind);
    return launch_conv(*op, in, res, out, B, mode, variant, nullptr, (hipStream_t)stream);
}

int romp_conv_num_variants(void) { return conv_num_variants(); }
int romp_conv_family_variants(int math) { return conv_family_variants(math); }

int romp_conv_trace_read(unsigned long long* dst_host, int max_words) { return conv_trace_read(dst_host, max_words); }

int romp_conv_describe(const romp_op* op, int B, int variant, char* out, int n) {
    ROMP_REQUIRE(op && out && n > 0 && B > 0, "romp_conv_describe: bad arguments");
    if (op->kind == ROMP_OP_STEM) { snprintf(out, n, "stem_conv"); return ROMP_OK; }
    if (op->kind == ROMP_OP_FUSESUM) { snprintf(out, n, "fusesum"); return ROMP_OK; }
    if (op->kind == ROMP_OP_FUSEUP) { snprintf(out, n, "fuseup"); return ROMP_OK; }
    if (op->kind == ROMP_OP_KSUM) { snprintf(out, n, "ksum"); return ROMP_OK; }
    if (op->kind == ROMP_OP_STEM7) { snprintf(out, n, "stem7_conv"); return ROMP_OK; }
    if (op->kind == ROMP_OP_MAXPOOL) { snprintf(out, n, "maxpool3s2"); return ROMP_OK; }
    if (op->kind == ROMP_OP_NOP) { snprintf(out, n, "nop"); return ROMP_OK; }
    if (op->kind == ROMP_OP_BBLOCK32) { snprintf(out, n, "bblock32"); return ROMP_OK; }
    if (op->kind == ROMP_OP_BBLOCK64) { snprintf(out, n, "bblock64"); return ROMP_OK; }
    if (op->kind == ROMP_OP_STEM2) { snprintf(out, n, "stem2"); return ROMP_OK; }
    if (op->kind == ROMP_OP_STEM7P) { snprintf(out, n, "stem7p"); return ROMP_OK; }
    if (op->kind == ROMP_OP_SEAM1X1) { snprintf(out, n, (op->flags & ROMP_OPF_SEAM_DS) ? "seam1x1_ds" : "seam1x1"); return ROMP_OK; }
    if (op->kind == ROMP_OP_FORK) { snprintf(out, n, "fork"); return ROMP_OK; }
    if (op->kind == ROMP_OP_JOIN) { snprintf(out, n, "join"); return ROMP_OK; }
    if (op->kind == ROMP_OP_RECORD) { snprintf(out, n, "record"); return ROMP_OK; }
    if (op->kind == ROMP_OP_WAIT) { snprintf(out, n, "wait"); return ROMP_OK; }
    if (op->kind == ROMP_OP_BEV_PACK) { snprintf(out, n, "bev_pack"); return ROMP_OK; }
    if (op->kind == ROMP_OP_BEV_MAPS) { snprintf(out, n, "bev_maps"); return ROMP_OK; }
    if (op->kind == ROMP_OP_CONV3D) { snprintf(out, n, "conv3d"); return ROMP_OK; }
    ROMP_REQUIRE(op->kind == ROMP_OP_CONV, "romp_conv_describe: op kind %d", op->kind);
    if (variant >= 0 && !conv_variant_valid(*op, variant)) { set_error("variant %d not valid for this op", variant); return ROMP_EINVAL; }
    return describe_conv(*op, B, variant, out, n);
}

float* romp_net_buffer_ptr(romp_net* n, int buf) {
    if (!n || buf < 0 || buf >= (int)n->bufs.size()) return nullptr;
    return n->bufs[buf];
}

int romp_net_tuned_variant(romp_net* n, int B, int op_index) {
    if (!n || op_index < 0 || op_index >= (int)n->ops.size()) return -1;
    const std::vector<int>* tv = tuned_for(n, B);
    return tv ? (*tv)[op_index] : -1;
}

// ---- plan files --------------------------------------------------------------------------------------------------
// The reference exports its network once (ROMPv1 -> ROMP.onnx, model.py:484-497) and its inference entry point then needs
// no model code (main.py:89,109).  The counterpart here: `romp_amd/export.py` writes the lowered program -- ops, arena
// sizes, packed constants, buffer initialisers, measured variant tables -- to one file; romp_net_load() builds a net from
// it with nothing but this library.  Layout (little endian, see export.py): PlanHeader, int64 buf_floats[n_bufs],
// romp_op ops[n_ops] whose pointer fields hold (offset + 1) into the device blob, or that with bit 63 set into the host
// blob (0 = null), PlanInit inits[n_inits], int32 tuned[n_tuned][2 + n_ops] (batch, n_variants at export, variant per op),
// device blob, host blob.
struct PlanHeader {
    char magic[8];                      // "ROMPPLAN"
    uint32_t version, abi, n_ops, n_bufs, n_inits, n_tuned, input_size, op_bytes;
    uint32_t split_k_items, flags;      // version 2: the KIND of plan -- > 0: single-image plan lowered with this work-item target; flags reserved (0)
    uint64_t center_floats, params_floats, dev_bytes, host_bytes;
};
struct PlanInit { int32_t buf; int32_t pad; uint64_t floats; uint64_t dev_off; };   // per-image content, replicated for every image

int romp_net_load(romp_net** out, const char* path, int max_batch) {
    ROMP_REQUIRE(out && path && max_batch > 0, "romp_net_load: bad arguments");
    FILE* f = fopen(path, "rb");
    ROMP_REQUIRE(f, "romp_net_load: cannot open %s", path);
    std::vector<char> file;
    {
        fseek(f, 0, SEEK_END);
        const long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        file.resize(sz > 0 ? (size_t)sz : 0);
        const size_t got = file.empty() ? 0 : fread(file.data(), 1, file.size(), f);
        fclose(f);
        ROMP_REQUIRE(got == file.size() && file.size() >= sizeof(PlanHeader), "romp_net_load: %s is truncated", path);
    }
    PlanHeader h;
    memcpy(&h, file.data(), sizeof(h));
    ROMP_REQUIRE(memcmp(h.magic, "ROMPPLAN", 8) == 0 && h.version == 2, "romp_net_load: %s is not a version-2 plan file", path);
    ROMP_REQUIRE(h.abi == ROMP_ABI_VERSION && h.op_bytes == sizeof(romp_op), "romp_net_load: plan was written for ABI %u (romp_op %u bytes), this library is ABI %d (%zu bytes)",
                 h.abi, h.op_bytes, ROMP_ABI_VERSION, sizeof(romp_op));
    size_t at = sizeof(PlanHeader);
    // every header field is bounded by the file size BEFORE it enters a product or a sum (a corrupted or crafted header must
    // not wrap the size check), then the sections must add up to the file exactly
    const uint64_t fsz = file.size();
    ROMP_REQUIRE(h.n_ops > 0 && h.n_ops <= fsz / sizeof(romp_op) && h.n_bufs <= fsz / 8 && h.n_inits <= fsz / sizeof(PlanInit) &&
                 h.dev_bytes <= fsz && h.host_bytes <= fsz && (h.n_tuned == 0 || h.n_tuned <= fsz / (4ull * (2 + (uint64_t)h.n_ops))),
                 "romp_net_load: %s: header fields exceed the file size", path);
    uint64_t need = at;
    const uint64_t parts[] = {(uint64_t)h.n_bufs * 8, (uint64_t)h.n_ops * sizeof(romp_op), (uint64_t)h.n_inits * sizeof(PlanInit),
                              (uint64_t)h.n_tuned * (2 + (uint64_t)h.n_ops) * 4, h.dev_bytes, h.host_bytes};
    bool overflow = false;
    for (uint64_t part : parts) overflow |= __builtin_add_overflow(need, part, &need);
    ROMP_REQUIRE(!overflow && need == fsz, "romp_net_load: %s: size %zu does not match its header", path, file.size());
    std::vector<int64_t> buf_floats(h.n_bufs);
    memcpy(buf_floats.data(), file.data() + at, (size_t)h.n_bufs * 8); at += (size_t)h.n_bufs * 8;
    std::vector<romp_op> ops(h.n_ops);
    memcpy(ops.data(), file.data() + at, (size_t)h.n_ops * sizeof(romp_op)); at += (size_t)h.n_ops * sizeof(romp_op);
    std::vector<PlanInit> inits(h.n_inits);
    if (h.n_inits) memcpy(inits.data(), file.data() + at, (size_t)h.n_inits * sizeof(PlanInit));
    at += (size_t)h.n_inits * sizeof(PlanInit);
    const int32_t* tuned = reinterpret_cast<const int32_t*>(file.data() + at); at += (size_t)h.n_tuned * (2 + h.n_ops) * 4;
    const char* dev_src = file.data() + at; at += h.dev_bytes;
    const char* host_src = file.data() + at;
    void* dev = nullptr;
    if (h.dev_bytes) {
        ROMP_HIP_CHECK(hipMalloc(&dev, h.dev_bytes));
        if (hipMemcpy(dev, dev_src, h.dev_bytes, hipMemcpyHostToDevice) != hipSuccess) { hipFree(dev); set_error("romp_net_load: upload failed"); return ROMP_EHIP; }
    }
    std::vector<char> host(host_src, host_src + h.host_bytes);
    bool bad = false;
    auto fix = [&](const void* field) -> const void* {
        const uint64_t v = (uint64_t)(uintptr_t)field;
        if (v == 0) return nullptr;
        const bool on_host = (v >> 63) != 0;
        const uint64_t off = (v & ~(1ull << 63)) - 1;
        if (off >= (on_host ? h.host_bytes : h.dev_bytes)) { bad = true; return nullptr; }
        return on_host ? (const void*)(host.data() + off) : (const void*)((const char*)dev + off);
    };
    for (romp_op& op : ops) {
        op.weight = (const float*)fix(op.weight); op.scale = (const float*)fix(op.scale); op.shift = (const float*)fix(op.shift);
        op.weight_aux = fix(op.weight_aux); op.weight_h2 = fix(op.weight_h2); op.scale_h2 = (const float*)fix(op.scale_h2);
    }
    if (bad) { if (dev) hipFree(dev); set_error("romp_net_load: %s: a constant lies outside its blob", path); return ROMP_EINVAL; }
    romp_net* n = nullptr;
    int rc = romp_net_create(&n, ops.data(), (int)h.n_ops, buf_floats.data(), (int)h.n_bufs, max_batch);
    if (rc) { if (dev) hipFree(dev); return rc; }
    n->plan_dev = dev;
    n->plan_host.swap(host);            // (vector swap keeps the storage address: the ops' host pointers stay valid)
    n->plan_input_size = (int32_t)h.input_size;
    n->plan_center_floats = (int64_t)h.center_floats;
    n->plan_params_floats = (int64_t)h.params_floats;
    n->plan_split_k_items = (int32_t)h.split_k_items;
    for (const PlanInit& in : inits) {
        if (in.buf < 0 || in.buf >= (int)h.n_bufs || buf_floats[in.buf] < 0 || in.floats > (uint64_t)buf_floats[in.buf] ||
            in.dev_off > h.dev_bytes || in.floats > (h.dev_bytes - in.dev_off) / 4) {      // (no sum that could wrap)
            set_error("romp_net_load: %s: bad buffer initialiser", path);
            romp_net_destroy(n);
            return ROMP_EINVAL;
        }
        for (int b = 0; b < max_batch; ++b)
            if (hipMemcpy(n->bufs[in.buf] + (size_t)b * buf_floats[in.buf], (const char*)dev + in.dev_off, in.floats * 4, hipMemcpyDeviceToDevice) != hipSuccess) {
                set_error("romp_net_load: buffer initialiser copy failed");
                romp_net_destroy(n);
                return ROMP_EHIP;
            }
    }
    for (uint32_t t = 0; t < h.n_tuned; ++t) {      // measured variant tables: only if this build still has the same kernels
        const int32_t* row = tuned + (size_t)t * (2 + h.n_ops);
        if (row[1] != conv_num_variants() || row[0] <= 0 || row[0] > max_batch) continue;
        if (romp_net_set_tuned(n, row[0], row + 2, (int)h.n_ops) != ROMP_OK) n->tuned.erase(row[0]);
    }
    *out = n;
    return ROMP_OK;
}

int romp_net_plan_info(romp_net* n, int32_t* input_size, int64_t* center_floats, int64_t* params_floats, int32_t* n_ops) {
    ROMP_REQUIRE(n, "romp_net_plan_info: null net");
    if (input_size) *input_size = n->plan_input_size;
    if (center_floats) *center_floats = n->plan_center_floats;
    if (params_floats) *params_floats = n->plan_params_floats;
    if (n_ops) *n_ops = (int32_t)n->ops.size();
    return ROMP_OK;
}

int romp_net_plan_kind(romp_net* n, int32_t* split_k_items) {
    ROMP_REQUIRE(n && split_k_items, "romp_net_plan_kind: bad arguments");
    *split_k_items = n->plan_split_k_items;
    return ROMP_OK;
}

int romp_net_set_tuned(romp_net* n, int B, const int32_t* variants, int n_ops) {
    ROMP_REQUIRE(n && variants && B > 0 && n_ops == (int)n->ops.size(), "romp_net_set_tuned: bad arguments");
    std::vector<int> tv(n_ops, -1);
    for (int i = 0; i < n_ops; ++i) {
        if (variants[i] < 0 || n->ops[i].kind != ROMP_OP_CONV) continue;
        ROMP_REQUIRE(conv_variant_valid(n->ops[i], variants[i]), "romp_net_set_tuned: variant %d is not valid for op %d", variants[i], i);
        tv[i] = variants[i];
    }
    n->tuned[B] = tv;
    drop_graphs(n);
    return ROMP_OK;
}

}  // extern "C"
